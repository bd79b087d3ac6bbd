#!/bin/bash
set -u
O=gpurun_out/r2q; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
GPS_GEMM_RING_MB=2 timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel" > $O/pytest_mb2.log 2>&1; echo "pytest mb2 rc=$?" >> $O/rc.txt
tail -3 $O/pytest_mb2.log
timeout 300 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids | sed 's/start.*prologue/prologue/; s/end \[.*//'
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring.txt 2>&1; cat $O/gemm_ring.txt | grep -v amdgpu.ids
GPS_GEMM_RING_MB=2 timeout 300 python tools/gemm_panel_bench.py 2>&1 | grep -E "sum:"
GPS_GEMM_RING_MB=1 timeout 300 python tools/gemm_panel_bench.py 2>&1 | grep -E "sum:"
cat $O/rc.txt

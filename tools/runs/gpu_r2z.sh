#!/bin/bash
set -u
O=gpurun_out/r2v; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc2.txt
tail -2 $O/pytest_gpu.log
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring.txt 2>&1
timeout 200 python tools/gemm_trace.py > $O/gemm_timeline.txt 2>&1
timeout 600 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
grep -v amdgpu $O/gemm_ring.txt | tail -14
cat $O/rc2.txt

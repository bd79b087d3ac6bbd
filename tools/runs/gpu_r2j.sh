#!/bin/bash
set -u
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel" -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "gemm_panel |passed|failed|Error" $O/pytest.log | tail -12
timeout 600 python tools/gemm_panel_bench.py > $O/gemm_panel.txt 2>&1; echo "gemm rc=$?" >> $O/rc.txt; cat $O/gemm_panel.txt | grep -v Tunable | tail -12
cat $O/rc.txt

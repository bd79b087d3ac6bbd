#!/bin/bash
set -u
O=gpurun_out/r2k; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel or gatedgcn" -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "gemm_panel |passed|failed|Error" $O/pytest.log | tail -12
timeout 600 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "fixture or baseline_sizes or ragged or custom_gnn" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
tail -3 $O/pytest_layer.log
GG_GRID=768:512,768:256,768:1024,384:1024 timeout 600 python tools/gg_sweep.py > $O/gg_sweep.txt 2>&1; cat $O/gg_sweep.txt
timeout 600 python tools/gemm_panel_bench.py > $O/gemm_panel.txt 2>&1; echo "gemm rc=$?" >> $O/rc.txt; cat $O/gemm_panel.txt | grep -v Tunable | tail -12
cat $O/rc.txt

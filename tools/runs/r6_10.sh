#!/bin/bash
# round 6: BN-backward fold: parity again, suite subset, then a kernel trace with the fold on
set -u
OUT=gpurun_out/r6_10; mkdir -p $OUT
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "bn_fold" -s > $OUT/fold.log 2>&1; echo "fold rc=$?"; grep -E "worst|passed|failed|Error" $OUT/fold.log | tail -8
python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py tests/test_hip_ops.py -q -x -m gpu -k "not favor" > $OUT/layer.log 2>&1; echo "layer rc=$?"; tail -3 $OUT/layer.log
bash tools/runs/r6_prof.sh $OUT > $OUT/prof.txt 2>&1
grep -E "k_gatedgcn_bwd|k_sattn_bwd|k_bwd_apply|k_bwd_partial|k_rows_fwd" $OUT/kernel_trace_stats_pcqm4m.txt | cut -c1-120
grep -n "k_sattn_bwd" -B3 -A3 $OUT/timeline_pcqm4m.txt | sed -n 1,40p | cut -c1-120

#!/bin/bash
set -u
OUT=gpurun_out/r6_11; mkdir -p $OUT
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "bn_fold" -s > $OUT/fold.log 2>&1; echo "fold rc=$?"; grep -E "worst|passed|failed|Error" $OUT/fold.log | tail -8
bash tools/runs/r6_ab.sh $OUT "fold3:" "fold2:GPS_GG_BN_FOLD=2" "fold1:GPS_GG_BN_FOLD=1" "fold0:GPS_GG_BN_FOLD=0"
python tools/kernel_probe.py 2>&1 | tail -2

#!/bin/bash
# round 6: BatchNorm-backward fold inside the GatedGCN backward: parity, then A/B
set -u
OUT=gpurun_out/r6_09; mkdir -p $OUT
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "bn_fold" -s > $OUT/fold.log 2>&1; echo "fold rc=$?"; grep -E "worst|passed|failed|Error" $OUT/fold.log | tail -8
python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py tests/test_hip_ops.py -q -x -m gpu -k "not favor" > $OUT/layer.log 2>&1; echo "layer rc=$?"; tail -3 $OUT/layer.log
bash tools/runs/r6_ab.sh $OUT "fold:" "nofold:GPS_GG_BN_FOLD=0"

#!/bin/bash
set -u
OUT=gpurun_out/r6_03; mkdir -p $OUT
python -m pytest tests/test_hip_norm.py -q -x -m gpu -k "gatedgcn_forward_emits" > $OUT/ggstats.log 2>&1; echo "ggstats rc=$?"; tail -n 3 $OUT/ggstats.log
python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py -q -x -m gpu > $OUT/model.log 2>&1; echo "model rc=$?"; tail -n 3 $OUT/model.log
bash tools/runs/r6_ab.sh $OUT "new:" "old:GPS_SMALL_LINEAR=0,GPS_MULTIHOT_WGRAD=0,GPS_GG_STATS=0" "no_small:GPS_SMALL_LINEAR=0" "no_mh:GPS_MULTIHOT_WGRAD=0" "no_gg:GPS_GG_STATS=0"

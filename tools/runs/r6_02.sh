#!/bin/bash
set -u
OUT=gpurun_out/r6_02; mkdir -p $OUT
python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "small_linear" > $OUT/small.log 2>&1; echo "small rc=$?"; tail -n 3 $OUT/small.log
python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py -q -x -m gpu -k "full_model or host_targets or fixture or code2_model" > $OUT/model.log 2>&1; echo "model rc=$?"; tail -n 3 $OUT/model.log
bash tools/runs/r6_ab.sh $OUT "new:" "old:GPS_SMALL_LINEAR=0,GPS_MULTIHOT_WGRAD=0" "small_only:GPS_MULTIHOT_WGRAD=0" "mh_only:GPS_SMALL_LINEAR=0"

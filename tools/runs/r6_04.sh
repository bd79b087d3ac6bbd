#!/bin/bash
set -u
OUT=gpurun_out/r6_04; mkdir -p $OUT
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "code2_model" -s > $OUT/code2.log 2>&1; echo "code2 rc=$?"; grep -E "code2 model|passed|failed" $OUT/code2.log | tail -n 6
python -m pytest tests/test_hip_ops.py tests/test_hip_norm.py -q -x -m gpu -k "gemm or wgrad or stats" > $OUT/gemm.log 2>&1; echo "gemm rc=$?"; tail -n 3 $OUT/gemm.log
bash tools/runs/r6_ab.sh $OUT "new:" "side:GPS_BLOCK_WGRAD_SIDE_STREAM=1" "fork0:GPS_CORE_FORK=0" "side_fork0:GPS_BLOCK_WGRAD_SIDE_STREAM=1,GPS_CORE_FORK=0"

#!/bin/bash
# persistent tiles (4-slot ring; pair dispatch + single-problem launches of >= 2 dispatch rounds): parity, then same-box A/Bs
set -u
OUT=gpurun_out/${1:-r6s4_persist5}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_norm.py -x -q -m gpu -k "gemm or persist" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
bash tools/runs/r6_ab_workload.sh $OUT/code2 code2 "persist:GPS_GEMM_SCHED=3" "no_persist:GPS_GEMM_SCHED=1"
bash tools/runs/r6_ab_workload.sh $OUT/pcqm pcqm4m "persist:GPS_GEMM_SCHED=3" "no_persist:GPS_GEMM_SCHED=1"

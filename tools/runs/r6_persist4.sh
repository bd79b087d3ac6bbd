#!/bin/bash
# persistent tiles on a 4-slot ring for the 128-column panels (code2's forward pair): parity, then the same-box A/B
set -u
OUT=gpurun_out/r6s4_persist4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gemm_panel_pair" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
bash tools/runs/r6_ab_workload.sh $OUT/code2 code2 "persist:GPS_GEMM_SCHED=3" "no_persist:GPS_GEMM_SCHED=1"

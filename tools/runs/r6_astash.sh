#!/bin/bash
# GatedGCN backward, per-node a_i stash (csrc/gatedgcn.hip ASTASH): parity tests, then same-box A/Bs of the code2 and pcqm4m steps.
set -u
OUT=gpurun_out/r6s4_astash; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gatedgcn" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
bash tools/runs/r6_ab_workload.sh $OUT/code2 code2 "astash:GPS_GG_ASTASH=1" "edge_stash:GPS_GG_ASTASH=0"
bash tools/runs/r6_ab_workload.sh $OUT/pcqm pcqm4m "astash:GPS_GG_ASTASH=1" "edge_stash:GPS_GG_ASTASH=0"

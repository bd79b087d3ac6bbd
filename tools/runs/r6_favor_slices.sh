#!/bin/bash
# FAVOR+ staged context kernels: slices per (graph, head) = what fills one dispatch round.  Parity, then the same-box A/B on code2.
set -u
OUT=gpurun_out/r6s4_favor3; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu -k "favor or performer or code2" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
bash tools/runs/r6_ab_workload.sh $OUT/code2 code2 "one_round:GPS_X=0" "slices4:GPS_FAVOR_SLICES=4" "slices2:GPS_FAVOR_SLICES=2"

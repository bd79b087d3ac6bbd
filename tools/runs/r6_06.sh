#!/bin/bash
# round 6: GINE block fix check, then the norm task-list launches alone: row blocks per task sweep (tools/norm_probe.py)
set -u
OUT=gpurun_out/r6_06; mkdir -p $OUT
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "gine" > $OUT/gine.log 2>&1; echo "gine rc=$?"; tail -2 $OUT/gine.log
for cfg in "256 256" "256 512" "256 1024" "256 2048" "512 1024" "384 1024"; do
  set -- $cfg
  GPS_NORM_BLOCKS=$1 GPS_NORM_FREE_BLOCKS=$2 python tools/norm_probe.py 2>&1 | tee -a $OUT/norm_probe.txt
done

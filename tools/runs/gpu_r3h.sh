#!/bin/bash
# round 3: what in the host-batch leg breaks a LATER capture? (record_stream bookkeeping vs cached blocks)
set -u
O=gpurun_out/r3h; mkdir -p $O
for cfg in "GPS_LOADER_RECORD_STREAM=0" "GPS_CAPTURE_EMPTY_CACHE=1" "GPS_LOADER_RECORD_STREAM=0 GPS_CAPTURE_EMPTY_CACHE=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "== $tag rc=$? $(grep -E "host-batch|launch-mode|timed region" $O/bench_$tag.err | tr '\n' '|')"
done

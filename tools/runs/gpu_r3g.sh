#!/bin/bash
# round 3: bisect of the capture segfault seen in r3f (after the host-batch leg / in step_cached)
set -u
O=gpurun_out/r3g; mkdir -p $O
for cfg in "NOLEG" "GPS_LOADER_BACKGROUND=0" "DEFAULT"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  extra=""; envs="$cfg"
  if [ "$cfg" = "NOLEG" ]; then extra="--no-h2d-leg"; envs="X=1"; fi
  if [ "$cfg" = "DEFAULT" ]; then envs="X=1"; fi
  env $envs timeout 300 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline $extra > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "== $tag rc=$? $(grep -E "host-batch|launch-mode|timed region" $O/bench_$tag.err | tr '\n' '|')"
  grep -A12 "Fatal Python" $O/bench_$tag.err | head -16
done
timeout 600 python -X faulthandler -m pytest tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "step_cached or device_loader or replay" > $O/pytest_optim.log 2>&1; echo "optim rc=$?"
tail -5 $O/pytest_optim.log; grep -A14 "Fatal Python" $O/pytest_optim.log | head -20

#!/bin/bash
# round 3: (1) does the purely linear capture (no tick, no branch stream) still fault now that no stale autograd graph
# reaches a capture?  (2) the whole GPU suite
set -u
O=gpurun_out/r3p; mkdir -p $O
for cfg in "GPS_CAPTURE_TICK=0 GPS_BRANCH_STREAM=0" "GPS_CAPTURE_TICK=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --launch graph > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "== [$cfg] rc=$? $(grep -E "launch-mode|timed region|captured|fault|Fault" $O/bench_$tag.err | tr '\n' '|')"
  tail -3 $O/bench_$tag.err | cut -c1-300
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"
tail -6 $O/pytest_gpu.log

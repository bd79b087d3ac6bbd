#!/bin/bash
# round 3: one weight-image launch + one BN-counter launch per STACK (GPS_STACK_PREP) A/B
set -u
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "not code2 and not graphormer and not san" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -4 $O/pytest.log
for i in 1 2; do
for cfg in "GPS_STACK_PREP=1" "GPS_STACK_PREP=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg 2>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],3), d.get('host_enqueue_ms_per_step'), d['launch_mode'], d['launch_trial_ms'])"
done
done

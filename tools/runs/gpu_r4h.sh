#!/bin/bash
# round 3: graph index built beside the encoders on a second stream (GPS_INDEX_PREFETCH) A/B
set -u
O=gpurun_out/r4h; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "full_model_train_step or hipgraph_replay_equals or step_cached or custom_gnn_vs_oracle" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for cfg in "GPS_INDEX_PREFETCH=1" "GPS_INDEX_PREFETCH=0" "GPS_INDEX_PREFETCH=1" "GPS_INDEX_PREFETCH=0"; do
  env $cfg timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['ms_per_step'],3), d['launch_mode'][:6], d['launch_trial_ms'])"
done

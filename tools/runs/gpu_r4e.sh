#!/bin/bash
# round 3: multi-hot embedding gradient on the split-K kernel (parity via the custom_gnn model test, A/B), code2 capture-after-eager test
set -u
O=gpurun_out/r4e; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_optim.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "eager_step_leaves or custom_gnn_vs_oracle or full_model_train_step" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -3 $O/pytest.log
for cfg in "GPS_MULTIHOT_WGRAD=1" "GPS_MULTIHOT_WGRAD=0" "GPS_MULTIHOT_WGRAD=1" "GPS_MULTIHOT_WGRAD=0"; do
  env $cfg timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['ms_per_step'],3), d['launch_mode'][:6], d['launch_trial_ms'])"
done

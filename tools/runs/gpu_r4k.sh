#!/bin/bash
# round 3: host-side diet of the block (raw stream query, cached module references): parity subset + host profile
set -u
O=gpurun_out/r4k; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "fixture or full_model_train_step or hipgraph_replay_equals or ragged or gradient_accumulation or dropout_on" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 200 python tools/host_profile.py --top 14 > $O/host_profile.txt 2>&1; echo rc=$?
sed -n 2,22p $O/host_profile.txt | cut -c1-140
for i in 1 2; do timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],3), d['launch_mode'][:6], 'host', round(d['host_enqueue_ms_per_step'],2), 'pcie', d.get('pcie_inclusive_ms_per_step'), d['launch_trial_ms'])"; done

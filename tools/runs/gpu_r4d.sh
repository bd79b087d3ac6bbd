#!/bin/bash
# round 3: GatedGCN core directly behind the merged projection (GPS_GG_FIRST) A/B; code2 bench line after the container scrub
set -u
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "not graphormer and not san and not signnet" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -3 $O/pytest.log
for i in 1 2; do
for cfg in "GPS_GG_FIRST=1" "GPS_GG_FIRST=0"; do
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d.get('in_step_kernel_ms',{})
def mean(sub):
    v=[x['ms']*1e3 for k,x in ks.items() if sub in k]; return round(sum(v)/len(v),1) if v else None
print('$cfg', round(d['ms_per_step'],3), d['launch_mode'][:6], 'frac', round(d['roofline']['frac'],3), 'gg_fwd', mean('k_gatedgcn_fwd'), 'sattn_fwd', mean('k_sattn_fwd'), 'rows_fwd', mean('k_rows_fwd'))"
done
done
timeout 900 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?"
tail -2 $O/bench_code2.err | cut -c1-200
python -c "
import json; d=json.loads(open('$O/bench_code2.json').read().strip().splitlines()[-1]); print('code2', round(d['ms_per_step'],3), round(d['value']), d['launch_mode'], d.get('pcie_inclusive_ms_per_step'), d['launch_trial_ms'])"

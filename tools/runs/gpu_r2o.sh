#!/bin/bash
set -u
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gatedgcn or gemm_panel" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -5 $O/pytest.log
timeout 1200 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -x > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
tail -6 $O/pytest_layer.log
GG_ONE=1 timeout 300 python tools/gg_sweep.py 2>&1 | grep RESULT
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2o/bench.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d.get('launch_trial_ms'), round(d['host_enqueue_ms_per_step'],2))
print(json.dumps(d['roofline']))
for k,v in d.get('kernel_rooflines',{}).items():
    print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('in_step_ms','isolated_hot_ms','isolated_rotating_ms','frac','frac_in_step')})
PY
cat $O/rc.txt

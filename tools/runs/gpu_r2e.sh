#!/bin/bash
set -u
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider --maxfail=12 > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?" > $O/rc.txt
tail -8 $O/pytest_ops.log
timeout 1500 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider --maxfail=12 -s > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
grep -E "passed|failed|FAILED|code2|pred  vs|parameter gradients" $O/pytest_layer.log | tail -12
timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench_pcqm4m.json 2> $O/bench_pcqm4m.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2e/bench_pcqm4m.json'))
print(d['ms_per_step'], d['value'], d['launch_mode'], d['launch_trial_ms'], d['host_enqueue_ms_per_step'])
print(d['roofline'])
for k,v in d['kernels'].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('in_step_ms','isolated_hot_ms','isolated_rotating_ms','frac','frac_isolated_hot','frac_isolated_rotating','mfma_frac')})
PY
cat $O/rc.txt

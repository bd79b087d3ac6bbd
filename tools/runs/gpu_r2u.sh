#!/bin/bash
set -u
O=gpurun_out/r2u; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "wgrad or accumulation" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -8 $O/pytest.log
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -x -k "fixture or baseline or oracle or code2" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
tail -4 $O/pytest_layer.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2u/bench.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d.get('launch_trial_ms'), round(d['host_enqueue_ms_per_step'],2))
w=d['kernels'].get('wgrad_grouped'); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in w.items() if k!='note'})
PY
cat $O/rc.txt

#!/bin/bash
set -u
O=gpurun_out/r2r; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -x > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" > $O/rc.txt
tail -4 $O/pytest_layer.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2r/bench.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d.get('launch_trial_ms'), round(d['host_enqueue_ms_per_step'],2))
k=d['in_step_kernel_ms']
tot=0
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms']*kv[1]['per_step'])[:14]:
    print(f"{v['ms']*v['per_step']:.3f} ms  {v['ms']*1e3:7.1f} us x {v['per_step']:.0f}  {n[:80]}")
PY
cat $O/rc.txt

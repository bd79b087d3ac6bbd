#!/bin/bash
set -u
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -x -k "fixture or baseline or determin" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" > $O/rc.txt
tail -3 $O/pytest_layer.log
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2y/bench.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d.get('launch_trial_ms'))
k=d['in_step_kernel_ms']
for n,v in k.items():
    if 'finalize' in n or 'k_rows' in n or 'k_bwd' in n: print(f"{v['ms']*1e3:7.1f} us x {v['per_step']:.0f}  {n[:60]}")
PY
cat $O/rc.txt

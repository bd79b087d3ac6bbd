#!/bin/bash
set -u
O=gpurun_out/r2x; mkdir -p $O
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-h2d-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" > $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2x/bench.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d.get('launch_trial_ms'))
for kn,v in d['kernels'].items():
    print(kn, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('in_step_ms','isolated_hot_ms','isolated_rotating_ms','frac')})
PY
timeout 200 python tools/gemm_panel_bench.py 2>&1 | grep -E "sum:|addend|relu|mask"

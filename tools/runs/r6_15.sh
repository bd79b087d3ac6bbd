#!/bin/bash
# round 6: full GPU suite on the tree after the hygiene cuts + norm probe (k_bwd_apply at 3 waves per SIMD) + bench line
set -u
OUT=gpurun_out/r6_15; mkdir -p $OUT
python -m pytest tests/ -q -m gpu -x > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
python tools/norm_probe.py 2>&1 | grep -v amdgpu | tee $OUT/norm_probe.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_15/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','launch_mode','dispatches_per_step')})
print({k:v for k,v in d.get('roofline',{}).items() if k in ('kernel','frac','launch_ms','step_share')}); print(d.get('roofline_step')); print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY

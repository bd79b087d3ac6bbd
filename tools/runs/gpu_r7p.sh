#!/bin/bash
# Round 5: after the staged-context threshold moved to long graphs: FAVOR / Performer / padding / loader tests, smoke, code2 + pcqm4m lines
set -u
O=gpurun_out/r7p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py tests/test_hip_padding.py -x -q -m gpu -k "favor or performer or Performer or code2" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
for w in code2 pcqm4m; do
timeout 200 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary > $O/bench_$w.json 2> $O/bench_$w.err
python - $O/bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]} {d['ms_per_step']:.3f} ms  {d['launch_mode'][:40]}")
PY
done

#!/bin/bash
set -u
O=gpurun_out/r3x; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  (cd _r2snap && timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $R/$O/bench_r2_$i.json 2> $R/$O/bench_r2_$i.err)
  for t in new r2; do echo "== $t $i: $(python -c "import json; d=json.loads(open('$O/bench_${t}_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d.get('launch_trial_ms'), d.get('pcie_inclusive_ms_per_step'))" 2>&1 | tail -1)"; done
done

#!/bin/bash
# round 3: dropout-ON parity + code2 attribution tests; same-box A/B of the step against the round-2 tree (_r2snap)
set -u
O=gpurun_out/r3d; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -s -k "dropout_on or code2_model" > $O/pytest.log 2>&1; echo "tests rc=$?" > $O/rc.txt
grep -E "passed|failed|dropout on|grad e max|parameter gradients|code2 model|per-graph" $O/pytest.log | tail -30
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  (cd _r2snap && timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $R/$O/bench_r2_$i.json 2> $R/$O/bench_r2_$i.err)
  for t in new r2; do echo "== $t $i: $(python -c "import json; d=json.loads(open('$O/bench_${t}_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('launch_trial_ms'))" 2>&1 | tail -1)"; done
done
cat $O/rc.txt

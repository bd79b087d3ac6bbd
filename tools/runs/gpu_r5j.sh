#!/bin/bash
# Round 4, fourth call (what is left of the budget): the optimizer / norm test files and the 10-layer masked-oracle test on
# the final tree (thread cap, salt fixture, grouped reference attention), then the bench line with the three-way loader leg.
set -u
O=gpurun_out/r5j; mkdir -p $O
T0=$(date +%s)
timeout 140 python -m pytest tests/test_hip_optim.py tests/test_hip_norm.py \
  "tests/test_hip_layer.py::test_full_model_with_dropout_on_vs_masked_oracle" \
  -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_sel.log 2>&1; echo "pytest_sel rc=$?" > $O/rc.txt
echo "t_sel=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_sel.log | tail -2
grep -n "^FAILED\|^ERROR\|^E  " $O/pytest_sel.log | head -20
grep -n "s call " $O/pytest_sel.log | head -10
timeout 120 python bench.py --bucketed-leg --no-cpu-baseline --no-kernel-roofline --steps 20 --warmup 5 > $O/bench_bucketed.json 2> $O/bench_bucketed.err; echo "bench rc=$?" >> $O/rc.txt
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5j/bench_bucketed.json').read().strip().splitlines()[-1])
    print('bench', round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('launch_trial_ms'), d.get('pcie_inclusive_ms_per_step'), json.dumps(d.get('pcie_inclusive_bucketed')))
except Exception as e: print('bench ERR', e)
PY
tail -4 $O/bench_bucketed.err
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt

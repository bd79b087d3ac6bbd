#!/bin/bash
# Round 4, second call: the new tests first (padded batches, eval-mode block; output kept with -s), then the full GPU suite
# serially with per-test durations, then the bench line with the bucketed loader leg.  Outputs under gpurun_out/r5h.
set -u
O=gpurun_out/r5h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 400 python -m pytest tests/test_hip_padding.py "tests/test_hip_norm.py::test_norm_lists_on_padded_batches_see_the_real_rows_only" \
  "tests/test_hip_norm.py::test_gemm_statistics_epilogue_skips_padding_rows" \
  "tests/test_hip_layer.py::test_eval_mode_block_matches_operator_path_and_oracle" \
  -m gpu -q -s -p no:cacheprovider --durations=20 > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" > $O/rc.txt
echo "t_new=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_new.log | tail -2
grep -n "^FAILED\|^ERROR\|padded vs un-padded\|bucketed stream" $O/pytest_new.log | head -30
timeout 560 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=40 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head -30
timeout 200 python bench.py --bucketed-leg --no-cpu-baseline > $O/bench_bucketed.json 2> $O/bench_bucketed.err; echo "bench rc=$?" >> $O/rc.txt
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5h/bench_bucketed.json').read().strip().splitlines()[-1])
    print('bench', round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('launch_trial_ms'), d.get('pcie_inclusive_ms_per_step'), json.dumps(d.get('pcie_inclusive_bucketed')))
except Exception as e: print('bench ERR', e)
PY
tail -3 $O/bench_bucketed.err
cat $O/rc.txt

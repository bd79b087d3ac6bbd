#!/bin/bash
# round 6: full GPU suite + driver-form bench line + kernel trace of the pcqm4m step, on commit 3142004
set -u
OUT=gpurun_out/r6_05; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/ -q -m gpu -x > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/gpu_suite.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
Q="--steps 90 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python bench.py $Q > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_05/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','launch_mode','dispatches_per_step')})
print(d.get('roofline')); print(d.get('roofline_step')); print(d.get('secondary'))
PY
head -40 $OUT/kernel_stats.csv | cut -c1-200

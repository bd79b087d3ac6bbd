#!/bin/bash
# round 6 A/B helper: quick bench legs (timed region only) under different switch settings, same box, interleaved twice.
#   usage: r6_ab.sh OUTDIR "NAME1:ENV1=..,ENV2=.." "NAME2:..." ...     (NAME:  = no extra environment)
set -u
OUT=$1; shift; mkdir -p $OUT
Q="--steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( for kv in ${envs//,/ }; do export "$kv"; done; python bench.py $Q > $OUT/${name}_$rep.json 2> $OUT/${name}_$rep.err )
    python - $OUT/${name}_$rep.json $name $rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], round(d['ms_per_step'],4), d['launch_mode'][:20], d.get('launch_trial_ms'))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
  done
done 2>&1 | tee $OUT/ab.txt

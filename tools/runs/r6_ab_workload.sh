#!/bin/bash
# A/B helper as r6_ab.sh for another bench workload:   r6_ab_workload.sh OUTDIR WORKLOAD "NAME:ENV=..,ENV=.." ...
set -u
OUT=$1; WL=$2; shift; shift; mkdir -p $OUT
Q="--workload $WL --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( for kv in ${envs//,/ }; do export "$kv"; done; python bench.py $Q > $OUT/${name}_$rep.json 2> $OUT/${name}_$rep.err )
    python - $OUT/${name}_$rep.json $name $rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], round(d['ms_per_step'],4))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
  done
done 2>&1 | tee $OUT/ab.txt

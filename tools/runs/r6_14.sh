#!/bin/bash
# round 6: split backward + ranged exchange on one rank (RCCL world 1), then bench with the forced exchange: split vs not
set -u
OUT=gpurun_out/r6_14; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_optim.py -q -x -m gpu -k "rccl_allreduce" > $OUT/t.log 2>&1; echo "test rc=$?"; tail -5 $OUT/t.log
Q="--steps 30 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
for v in 1 0; do
  GPS_BENCH_FORCE_REDUCER=1 GPS_DP_SPLIT=$v timeout 600 python bench.py $Q > $OUT/dp_split$v.json 2> $OUT/dp_split$v.err; echo "split=$v rc=$?"
  python - $OUT/dp_split$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['launch_mode'], d.get('final_loss'))
PY
done

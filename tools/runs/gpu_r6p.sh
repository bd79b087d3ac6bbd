#!/bin/bash
# Round 5: a last sweep of existing switches on the final tree (replayed step, same box).
set -u
O=gpurun_out/r6p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:20s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run default A=1
run norm128 GPS_NORM_BLOCKS=128
run norm512 GPS_NORM_BLOCKS=512
run ggstats GPS_GG_STATS=1
run ggfirst GPS_GG_FIRST=1 GPS_CORE_FORK=0
run nofork GPS_CORE_FORK=0
run default2 A=1

#!/bin/bash
# Round 5: zinc secondary (launch-latency-bound) -- is 2.87 vs 2.43 ms the box or the 16-byte weight-gradient reduce?
set -u
O=gpurun_out/r6z; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --workload zinc --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:20s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run vec A=1
run scalar GPS_WGRAD_REDUCE_VEC=0
run vec2 A=1
run scalar2 GPS_WGRAD_REDUCE_VEC=0
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run pcqm4m A=1

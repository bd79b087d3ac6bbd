#!/bin/bash
# Round 5: zinc replayed step, yesterday's tree (aa30021, unpacked under _old_tree) against today's, same box, interleaved
set -u
O=$(pwd)/gpurun_out/r6z2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
A="--workload zinc --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; dir=$2
  ( cd $dir && timeout 200 python bench.py $A > $O/bench_$n.json 2> $O/bench_$n.err )
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:20s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run new1 .
run old1 _old_tree
run new2 .
run old2 _old_tree

#!/bin/bash
# Round 5, call 9: HIP runtime knobs that touch hipGraph replay / kernel boundaries, A/B on the replayed step
# (335 dispatches per step: a boundary costs 1.5-2 us each).
set -u
O=gpurun_out/r6i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  host {d['host_enqueue_ms_per_step']:.2f} ms  loss {d['final_loss']:.6f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run default A=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run skiprelease DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run graphqueues DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run default2 A=1

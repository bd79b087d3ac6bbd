#!/bin/bash
# Round 5, call 10: two (graph, head) items per wavefront in the attention forward (GPS_SATTN_FWD2=1): parity at the sizes
# that take it (dh = 24, 4,096 items), then the step A/B and the kernel's own time.
set -u
O=gpurun_out/r6j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPS_SATTN_FWD2=1 timeout 400 python -m pytest tests/test_hip_layer.py -q -p no:cacheprovider -x -k "baseline_sizes or fused_block_with_dropout or full_model_with_dropout or full_model_vs_oracle" > $O/pytest_fwd2.log 2>&1; rc=$?; echo "pytest fwd2 rc=$rc"
tail -3 $O/pytest_fwd2.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 300 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get('in_step_kernel_ms',{})
    fw=[(kk,v) for kk,v in k.items() if 'k_sattn_fwd' in kk]
    print(f"{sys.argv[2]:20s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}  ", [(kk[28:52], round(v['ms']*1e3,1)) for kk,v in fw])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run default A=1
run fwd2 GPS_SATTN_FWD2=1
run default2 A=1
run fwd2b GPS_SATTN_FWD2=1

#!/bin/bash
# Round 5, call 11: paired column slots in the block-form attention kernels (one 8-byte load feeds both dh tiles): parity
# (op-level attention tests at every head width, layer tests), then a same-box A/B against a library carrying the previous
# attention objects (tools/micro/abl_attn/old_attention.so), with the kernels' in-step and isolated durations.
set -u
O=gpurun_out/r6k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -q -p no:cacheprovider -x -k "attention or attn or baseline_sizes or fused_block_with_dropout or full_model or fixture" > $O/pytest_attn.log 2>&1; rc=$?; echo "pytest attn rc=$rc"
tail -3 $O/pytest_attn.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 300 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d['kernels']
    f,b=k['seg_attn_fwd'],k['seg_attn_bwd']
    print(f"{sys.argv[2]:14s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}  fwd in-step {f['ms']*1e3:.1f} hot {f['isolated_hot_ms']*1e3:.1f} rot {f['isolated_rotating_ms']*1e3:.1f} | bwd in-step {b['ms']*1e3:.1f} hot {b['isolated_hot_ms']*1e3:.1f} rot {b['isolated_rotating_ms']*1e3:.1f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run new A=1
run old GPS_HIP_LIB=$PWD/tools/micro/abl_attn/old_attention.so
run new2 A=1
run old2 GPS_HIP_LIB=$PWD/tools/micro/abl_attn/old_attention.so

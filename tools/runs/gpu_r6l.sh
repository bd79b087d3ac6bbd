#!/bin/bash
# Round 5, call 12: FAVOR+ with the four dh tiles' column-form operand elements in ONE 16-byte load (column slots): parity
# (op tests, reference fixture, Performer block with dropout, code2 models), then code2 A/B against the previous favor object.
set -u
O=gpurun_out/r6l; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py tests/test_hip_padding.py -q -p no:cacheprovider -x -k "favor or performer or code2" > $O/pytest_favor.log 2>&1; rc=$?; echo "pytest favor rc=$rc"
tail -3 $O/pytest_favor.log
B="python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 300 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get('in_step_kernel_ms',{})
    fv={kk.split('k_favor_')[1].split('(')[0]: round(v['ms']*1e3,1) for kk,v in k.items() if 'k_favor_' in kk}
    print(f"{sys.argv[2]:8s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}  favor us: {fv}  sum {sum(fv.values()):.0f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run new A=1
run old GPS_HIP_LIB=$PWD/tools/micro/abl_favor/old_favor.so
run new2 A=1
run old2 GPS_HIP_LIB=$PWD/tools/micro/abl_favor/old_favor.so

#!/bin/bash
# Round 5, call 3: the fp16-form ring GEMM with loader wavefronts (k_gemm_ring16L): parity first (every GEMM / statistics /
# norm test), then stand-alone timing against the round-4 kernel, then the step A/B.
set -u
O=gpurun_out/r6c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 400 python -m pytest tests/test_hip_ops.py tests/test_hip_norm.py -q -p no:cacheprovider -x -k "gemm or wgrad or stat or norm or race" > $O/pytest_gemm.log 2>&1; rc=$?; echo "pytest gemm rc=$rc"
tail -5 $O/pytest_gemm.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" $O/pytest_gemm.log | head -20; fi
echo "t_tests=$(( $(date +%s) - T0 ))"
for l in 0 1; do GPS_GEMM_LOADERS=$l timeout 120 python tools/ring_ablate_bench.py 2>/dev/null | tail -1 | sed "s/^default/loaders=$l/"; done | tee $O/ring_loaders.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.4f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
if [ $rc -eq 0 ]; then
run loaders A=1
run noloaders GPS_GEMM_LOADERS=0
run loaders_corefork GPS_CORE_FORK=1
run loaders2 A=1
echo "t_ab=$(( $(date +%s) - T0 ))"
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_hip_ops.py::test_gemm_panel_split_products > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -4 $O/pytest_all.log
fi
echo "t_all=$(( $(date +%s) - T0 ))"

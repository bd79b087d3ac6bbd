#!/bin/bash
# Round 5: weight-gradient reduce -- four outputs per thread, and folded into the last-arriving slice (gps_wgrad_grouped_sync).
set -u
O=gpurun_out/r6q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wgrad or race_screen" > $O/tests_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/tests_ops.log
timeout 900 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py tests/test_hip_padding.py -x -q -m gpu > $O/tests_block.log 2>&1; echo "block rc=$?"; tail -3 $O/tests_block.log
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
run fold A=1
run nofold_vec GPS_WGRAD_FOLD=0
run nofold_scalar GPS_WGRAD_FOLD=0 GPS_WGRAD_REDUCE_VEC=0
run fold2 A=1
run nofold_vec2 GPS_WGRAD_FOLD=0
run nofold_scalar2 GPS_WGRAD_FOLD=0 GPS_WGRAD_REDUCE_VEC=0
cd /tmp; export TMPDIR=/tmp
for m in fold nofold; do
  F=1; [ $m = nofold ] && F=0
  rm -rf /tmp/kt_$m
  GPS_WGRAD_FOLD=$F timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$m -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph > $GRAFT_REPO_ROOT/$O/kt_$m.log 2>&1
  S=$(find /tmp/kt_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && grep -i "wgrad" $S > $GRAFT_REPO_ROOT/$O/kt_${m}_wgrad.csv
done
cd $GRAFT_REPO_ROOT; cat $O/kt_*_wgrad.csv

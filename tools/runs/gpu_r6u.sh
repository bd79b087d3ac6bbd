#!/bin/bash
# Round 5: staging thread with its host operators on one thread + memcpy into the pinned ring: probe, loader / padding tests;
# does the OpenMP wait policy move the eager step (spinning intra-op workers beside the launching thread)?
set -u
O=gpurun_out/r6u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/loader_stage_probe.py > $O/probe.txt 2> $O/probe.err; echo rc=$?
cat $O/probe.txt
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_hip_padding.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  {d['launch_mode'][:30]}  trial {d.get('launch_trial_ms')}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run auto_default A=1
run auto_passive OMP_WAIT_POLICY=PASSIVE
B="$B --launch eager"
run eager_default A=1
run eager_passive OMP_WAIT_POLICY=PASSIVE
run eager_spin0 GOMP_SPINCOUNT=0
run eager_default2 A=1

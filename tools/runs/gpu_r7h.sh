#!/bin/bash
# Round 5: parallel chunk prefix in the FAVOR+ chunked kernels: identity test (incl. 300 graphs), tools check, code2 step
set -u
O=gpurun_out/r7h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -x -q -m gpu -k "favor or performer or Performer" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check rc=$?"; grep -v "vs plain" $O/check.txt
timeout 200 python bench.py --workload code2 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph > $O/bench_code2.json 2> $O/bench_code2.err
python - $O/bench_code2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"code2 {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.8f}")
PY

#!/bin/bash
# round 4, call 4: the whole GPU suite on the current tree (records of 8 words, pooled records, GatedGCN-made records,
# hardened trees, new parity tests) + the step with and without the fp16 forms
set -u
O=gpurun_out/r5d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|error" $O/pytest_gpu.log | tail -5
grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
for cfg in "1 1" "0 0"; do
  set -- $cfg
  GPS_GEMM_F16=$1 GPS_WGRAD_F16=$2 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_$1$2.json 2> $O/bench_$1$2.err; echo "bench gemm16=$1 wgrad16=$2 rc=$?"
  python -c "
import json; d=json.loads(open('$O/bench_$1$2.json').read().strip().splitlines()[-1]); print('GEMM_F16=$1 WGRAD_F16=$2', round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))"
done
timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_code2.json').read().strip().splitlines()[-1]); print('code2', round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))"

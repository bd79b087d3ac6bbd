#!/bin/bash
# round 4, call 1: fp16 (3-product) ring GEMM -- parity of both forms, the GEMM table, the step A/B
set -u
O=gpurun_out/r5a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "gemm_panel_fp32 or gemm16 or race_screen" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -5 $O/pytest_gemm.log
timeout 600 python -m pytest tests/test_hip_norm.py -x -q -k "gemm_epilogue" > $O/pytest_stats.log 2>&1; echo "pytest stats rc=$?"; tail -3 $O/pytest_stats.log
timeout 600 python tools/gemm_panel_bench.py > $O/gemm_table.txt 2> $O/gemm_table.err; echo "table rc=$?"; cat $O/gemm_table.txt
for f in 1 0; do
  GPS_GEMM_F16=$f timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_f16_$f.json 2> $O/bench_f16_$f.err; echo "bench f16=$f rc=$?"
  python -c "
import json; d=json.loads(open('$O/bench_f16_$f.json').read().strip().splitlines()[-1]); print('F16=$f', round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))"
done

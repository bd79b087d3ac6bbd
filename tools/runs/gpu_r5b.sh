#!/bin/bash
# round 4, call 2: absmax fixed (one atomic per workgroup), words made by producers (ring epilogue, norm tasks), wgrad fp16 form
set -u
O=gpurun_out/r5b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "gemm_panel_fp32 or gemm16 or race_screen or wgrad" > $O/pytest_gemm.log 2>&1; echo "pytest gemm/wgrad rc=$?"; tail -5 $O/pytest_gemm.log
timeout 900 python -m pytest tests/test_hip_norm.py tests/test_hip_layer.py -x -q > $O/pytest_layer.log 2>&1; echo "pytest norm+layer rc=$?"; tail -8 $O/pytest_layer.log
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_table.txt 2> $O/gemm_table.err; echo "table rc=$?"; head -12 $O/gemm_table.txt
timeout 300 python tools/wgrad16_probe.py > $O/wgrad16.txt 2> $O/wgrad16.err; echo "wgrad probe rc=$?"; cat $O/wgrad16.txt
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  GPS_GEMM_F16=$1 GPS_WGRAD_F16=$2 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_$1$2.json 2> $O/bench_$1$2.err; echo "bench gemm16=$1 wgrad16=$2 rc=$?"
  python -c "
import json; d=json.loads(open('$O/bench_$1$2.json').read().strip().splitlines()[-1]); print('GEMM_F16=$1 WGRAD_F16=$2', round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))"
done

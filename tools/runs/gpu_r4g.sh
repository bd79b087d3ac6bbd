#!/bin/bash
# round 3: mixed tile heights in the merged projection's launch (GPS_GEMM_TAIL) -- parity, race screen, A/B
set -u
O=gpurun_out/r4g; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "(gemm_panel_fp32 and (7569-384-2688 or 7569-256-1792 or 25013-256-256 or 15348-384-384 or 2000-608-1216)) or dma_kernels_race" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/pytest_ops.log
timeout 200 python tools/gemm_panel_bench.py 2>/dev/null | sed -n 2,4p
GPS_GEMM_TAIL=0 timeout 200 python tools/gemm_panel_bench.py 2>/dev/null | sed -n 3,3p
for cfg in "GPS_GEMM_TAIL=1" "GPS_GEMM_TAIL=0" "GPS_GEMM_TAIL=1" "GPS_GEMM_TAIL=0"; do
  env $cfg timeout 200 python bench.py --steps 30 --warmup 10 --launch graph --no-cpu-baseline --no-h2d-leg --no-kernel-roofline 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['ms_per_step'],3), d['launch_mode'][:6])"
done

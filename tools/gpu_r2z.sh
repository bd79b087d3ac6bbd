#!/bin/bash
set -u
O=gpurun_out/r2z; mkdir -p $O
for i in 1 2 3; do
for m in 1 0; do
GPS_GEMM_MERGE=$m timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline --launch graph > $O/bench_m$m.json 2> $O/bench_m$m.err
python - $m <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2z/bench_m{sys.argv[1]}.json'))
print('merge', sys.argv[1], round(d['ms_per_step'],3), round(d['value']))
PY
done
done

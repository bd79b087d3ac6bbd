#!/bin/bash
set -u
O=gpurun_out/r2z; mkdir -p $O
for rep in 1 2; do
for cfg in "1 768" "0 768" "0 512" "1 512"; do
set -- $cfg
GPS_BRANCH_STREAM=$1 GPS_GG_FWD_THREADS=$2 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-h2d-leg --launch graph > $O/b.json 2> $O/b.err
python - $1 $2 <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2z/b.json'))
k=d['kernels']
print('branch',sys.argv[1],'fwdthreads',sys.argv[2], round(d['ms_per_step'],3), 'gg_fwd in-step us', round(k['gatedgcn_fwd']['in_step_ms']*1e3,1), 'gg_bwd', round(k['gatedgcn_bwd']['in_step_ms']*1e3,1), 'attn fwd', round(k['seg_attn_fwd']['in_step_ms']*1e3,1), 'bwd', round(k['seg_attn_bwd']['in_step_ms']*1e3,1))
PY
done
done

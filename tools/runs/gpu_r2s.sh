#!/bin/bash
set -u
O=gpurun_out/r2s; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2s/bench_{sys.argv[1]}.json'))
print(sys.argv[1], round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:12], d.get('launch_trial_ms'), round(d['host_enqueue_ms_per_step'],2))
PY
}
run side1 GPS_WGRAD_SIDE_STREAM=1
run side0 GPS_WGRAD_SIDE_STREAM=0
run side0_noring GPS_WGRAD_SIDE_STREAM=0 GPS_GEMM_RING=0
run side1_noring GPS_WGRAD_SIDE_STREAM=1 GPS_GEMM_RING=0

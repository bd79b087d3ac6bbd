#!/bin/bash
set -u
O=gpurun_out/r2q; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring.txt 2>&1; cat $O/gemm_ring.txt | grep -v amdgpu.ids
GPS_WGRAD_SIDE_STREAM=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $O/bench_side0.json 2> $O/bench_side0.err
GPS_WGRAD_SIDE_STREAM=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $O/bench_side1.json 2> $O/bench_side1.err
python - <<'PY'
import json
for n in ('side0','side1'):
    d=json.load(open(f'gpurun_out/r2q/bench_{n}.json'))
    print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:12], d.get('launch_trial_ms'), round(d['host_enqueue_ms_per_step'],2))
PY
cat $O/rc.txt

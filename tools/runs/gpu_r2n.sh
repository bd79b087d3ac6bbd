#!/bin/bash
set -u
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel or edge_attention" -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "offset|passed|failed|Error|error" $O/pytest.log | tail -10
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "san or fixture or baseline" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
tail -6 $O/pytest_layer.log
timeout 300 python tools/gemm_panel_bench.py 2>&1 | grep -E "split_weights|sum:" 
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_panel.json 2> $O/bench_panel.err; echo "bench rc=$?" >> $O/rc.txt
GPS_GEMM_PANEL=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_nopanel.json 2> $O/bench_nopanel.err; echo "bench2 rc=$?" >> $O/rc.txt
python - <<'PY'
import json
for f in ('bench_panel','bench_nopanel'):
    d=json.load(open(f'gpurun_out/r2n/{f}.json'))
    print(f, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:20], d['launch_trial_ms'], round(d['host_enqueue_ms_per_step'],2))
PY
cat $O/rc.txt

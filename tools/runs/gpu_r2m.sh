#!/bin/bash
set -u
O=gpurun_out/r2m; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel" -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "offset|passed|failed|Error" $O/pytest.log | tail -10
timeout 1500 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider --maxfail=12 > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?" >> $O/rc.txt
tail -6 $O/pytest_layer.log
timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench_pcqm4m.json 2> $O/bench_pcqm4m.err; echo "bench rc=$?" >> $O/rc.txt
GPS_GEMM_PANEL=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_nopanel.json 2> $O/bench_nopanel.err; echo "bench2 rc=$?" >> $O/rc.txt
python - <<'PY'
import json
for f in ('bench_pcqm4m','bench_nopanel'):
    d=json.load(open(f'gpurun_out/r2m/{f}.json'))
    print(f, d['ms_per_step'], d['value'], d['launch_mode'][:20], d['launch_trial_ms'], d['host_enqueue_ms_per_step'])
d=json.load(open('gpurun_out/r2m/bench_pcqm4m.json'))
for k,v in d['kernels'].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('in_step_ms','isolated_hot_ms','isolated_rotating_ms','frac','mfma_frac')})
for k,v in list(d['in_step_kernel_ms'].items())[:16]:
    print(round(v['ms']*1000,1), round(v['per_step'],1), k[:100])
PY
cat $O/rc.txt

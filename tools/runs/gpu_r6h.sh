#!/bin/bash
# Round 5, call 8: the N > 1 control flow of bench.py on the 1-GPU box -- two ranks sharing device 0, collectives on gloo
# (GPS_BENCH_SHARE_GPU=1 GPS_BENCH_BACKEND=gloo: a test mode, not a measurement) -- through the driver's command line, with
# the kernel-roofline pass on (the pass that held a rank-0-only collective until this round); then the default line once more.
set -u
O=gpurun_out/r6h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
GPS_BENCH_SHARE_GPU=1 GPS_BENCH_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"
grep -n "launch-mode\|timed region\|captured\|broadcast\|Traceback\|Error" $O/bench_2ranks.err | head -20
python - <<'PY'
import json
try:
    lines=[l for l in open('gpurun_out/r6h/bench_2ranks.json').read().splitlines() if l.startswith('{')]
    print('json lines:', len(lines))
    d=json.loads(lines[-1]); print({k:d[k] for k in ('n_gpus','ms_per_step','value','launch_mode','collective_backend','grad_allreduce_bytes')}); print(d['config']['parallelism'], d['launch_trial_ms'] and {k:v for k,v in d['launch_trial_ms'].items() if k!='rounds'}); print('roofline' in d, 'kernels' in d)
except Exception as e: print('ERR', e)
PY
echo "t_2rank=$(( $(date +%s) - T0 ))"
GPS_BENCH_SHARE_GPU=1 GPS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 2 --steps 10 --warmup 3 --exchange bucketed --no-kernel-roofline > $O/bench_2ranks_bucketed.json 2> $O/bench_2ranks_bucketed.err; echo "2-rank bucketed rc=$?"
tail -2 $O/bench_2ranks_bucketed.err | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -n "launch-mode trial\|timed region\|secondary" $O/bench_default.err
echo "t_all=$(( $(date +%s) - T0 ))"

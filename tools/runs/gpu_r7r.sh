#!/bin/bash
# Round 5: the N > 1 control flow of bench.py once more on the last commit (two ranks sharing device 0, collectives on gloo: a
# test mode, not a measurement), through the driver's command line
set -u
O=gpurun_out/r7r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPS_BENCH_SHARE_GPU=1 GPS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 10 --warmup 3 --no-kernel-roofline > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"
grep -n "launch-mode\|timed region\|captured\|broadcast\|Traceback\|Error" $O/bench_2ranks.err | head -12
python - <<'PY'
import json
try:
    lines=[l for l in open('gpurun_out/r7r/bench_2ranks.json').read().splitlines() if l.startswith('{')]
    print('json lines:', len(lines))
    d=json.loads(lines[-1]); print({k:d[k] for k in ('n_gpus','ms_per_step','value','launch_mode','collective_backend','grad_allreduce_bytes','scaling')})
except Exception as e: print('ERR', e)
PY

#!/bin/bash
# round 3: the bench's data-parallel path (RCCL, world size 1 forced) and the launcher path (torch.distributed.run, 1 rank)
set -u
O=gpurun_out/r4i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPS_BENCH_FORCE_REDUCER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_reducer.json 2> $O/bench_reducer.err; echo "forced reducer rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_reducer.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['launch_mode'], d['n_gpus'], d['scaling'], d['config'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_torchrun.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['launch_mode'], d['n_gpus'])"

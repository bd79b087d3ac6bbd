#!/bin/bash
# Round 5, call 14: the bn_node_x apply moved in front of the backward fork (it ran 33 us instead of 11 beside the attention
# backward and held the GatedGCN backward back): parity of the block, step A/B (GPS_CORE_FORK=0 / bwd / 1), timeline.
set -u
O=gpurun_out/r6n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py -q -p no:cacheprovider -x -k "fused_block or full_model_with_dropout or baseline_sizes or invisible_to_the_real or performer_block" > $O/pytest_blk.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_blk.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:20s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.6f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run fork_bwd A=1
run fork_none GPS_CORE_FORK=0
run fork_both GPS_CORE_FORK=1
run fork_bwd2 A=1
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_n
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning --no-secondary --launch graph > $R/$O/prof.json 2> $R/$O/prof.log
DB=$(find /tmp/prof_n -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline.txt 2>&1
cd $R; sed -n 172,182p $O/timeline.txt | cut -c1-90

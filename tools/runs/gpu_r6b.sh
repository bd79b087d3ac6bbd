#!/bin/bash
# Round 5, call 2: (1) ablation builds of the fp16-form ring GEMM (what the loop spends its time on), (2) the narrow fork --
# attention core beside the GatedGCN core -- A/B in the replayed step with the GatedGCN backward's stash sized so that both
# backward kernels fit a CU, GatedGCN launch-shape variants, the multi-hot gradient switch, (3) parity of the forked block,
# (4) the padding tests the first call stopped in front of.
set -u
O=gpurun_out/r6b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
for v in base nodma nosplit nord nobar nostore nomfma nodma_nosplit onlymfma; do
  GPS_HIP_LIB=$PWD/tools/micro/abl_ring/$v.so timeout 120 python tools/ring_ablate_bench.py 2>/dev/null | tail -1
done | tee $O/ring_ablate.txt
echo "t_abl=$(( $(date +%s) - T0 ))"
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.4f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run default A=1
run corefork GPS_CORE_FORK=1
run corefork_stash80 GPS_CORE_FORK=1 GPS_GG_STASH_KB=80
run corefork_stash64 GPS_CORE_FORK=1 GPS_GG_STASH_KB=64
run stash80 GPS_GG_STASH_KB=80
run wg1024_stash48 GPS_GG_TARGET_WG=1024 GPS_GG_STASH_KB=48
run multihot GPS_MULTIHOT_WGRAD=1
run default2 A=1
echo "t_ab=$(( $(date +%s) - T0 ))"
GPS_CORE_FORK=1 GPS_GG_STASH_KB=80 timeout 300 python -m pytest tests/test_hip_layer.py -q -p no:cacheprovider -x -k "fused_block or full_model_with_dropout or gpslayer_vs_oracle_baseline" > $O/pytest_corefork.log 2>&1; echo "pytest corefork rc=$?"
tail -3 $O/pytest_corefork.log
timeout 300 python -m pytest tests/test_hip_padding.py tests/test_hip_optim.py -q -p no:cacheprovider -s > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?"
grep -n "passed\|failed\|replayed vs eager\|graphs reversed\|padded vs un-padded\|ff_linear1" $O/pytest_rest.log | head -20
export TMPDIR=/tmp; R=$PWD
cd /tmp; rm -rf /tmp/prof_cf
GPS_CORE_FORK=1 GPS_GG_STASH_KB=80 timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_cf -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning --no-secondary --launch graph > $R/$O/prof_cf.json 2> $R/$O/prof_cf.log
DB=$(find /tmp/prof_cf -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 30 > $R/$O/kernel_trace_stats_corefork.txt 2>&1 && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_corefork.txt 2>&1
cd $R
echo "t_all=$(( $(date +%s) - T0 ))"

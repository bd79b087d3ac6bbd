#!/bin/bash
# round 4, call 6: records spread over eight cache lines (attention regression of call 5), FAVOR+ A/B of unconditional loads
set -u
O=gpurun_out/r5f; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "attention or attn or gemm16 or wgrad_split" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -3 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_hip_layer.py -x -q -k "fused_block_with_dropout or full_model_train_step or fixture" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?"; tail -3 $O/pytest_layer.log
GPS_HIP_LIB=$R/graphgps_amd/csrc/libgps_hip_favor_uncond.so timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -x -q -k "favor or performer or Performer" > $O/pytest_favor_uncond.log 2>&1; echo "pytest favor (uncond lib) rc=$?"; tail -3 $O/pytest_favor_uncond.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_pcqm4m.json 2> $O/bench_pcqm4m.err; echo "bench pcqm4m rc=$?"
timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?"
GPS_HIP_LIB=$R/graphgps_amd/csrc/libgps_hip_favor_uncond.so timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2_uncond.json 2> $O/bench_code2_uncond.err; echo "bench code2 (uncond lib) rc=$?"
python - <<'PY'
import json
for n in ('bench_pcqm4m','bench_code2','bench_code2_uncond'):
    try:
        d=json.loads(open(f'gpurun_out/r5f/{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))
    except Exception as e: print(n,'ERR',e)
PY
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_p
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_pcqm4m.json 2> $R/$O/prof_pcqm4m.log
DB=$(find /tmp/prof_p -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_pcqm4m.txt 2>&1
rm -rf /tmp/prof_p /tmp/prof_c
GPS_HIP_LIB=$R/graphgps_amd/csrc/libgps_hip_favor_uncond.so timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o bench -- python $R/bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_code2_uncond.json 2> $R/$O/prof_code2_uncond.log
DB=$(find /tmp/prof_c -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 30 > $R/$O/kernel_trace_stats_code2_uncond.txt 2>&1
rm -rf /tmp/prof_c
cd $R
head -24 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150
head -12 $O/kernel_trace_stats_code2_uncond.txt | cut -c1-150

#!/bin/bash
# round 4, call 5: attention with the max|.| records and K^T in LDS, FAVOR+ row slices, relaxed-bar tests; benches + traces
set -u
O=gpurun_out/r5e; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "attention or attn or favor or gatedgcn or gemm16" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -3 $O/pytest_ops.log
timeout 1500 python -m pytest tests/test_hip_layer.py -x -q -k "dropout_on or performer or baseline_sizes or fixture or code2 or ragged" > $O/pytest_layer.log 2>&1; echo "pytest layer rc=$?"; tail -4 $O/pytest_layer.log
grep -n "Performer block\|parameter gradients\|pred vs fp64\|three-way" $O/pytest_layer.log | head
for w in pcqm4m code2; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
GPS_FAVOR_SLICES=1 timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2_s1.json 2> $O/bench_code2_s1.err; echo "bench code2 slices=1 rc=$?"
python - <<'PY'
import json
for n in ('bench_pcqm4m','bench_code2','bench_code2_s1'):
    try:
        d=json.loads(open(f'gpurun_out/r5e/{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))
    except Exception as e: print(n,'ERR',e)
PY
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
head -28 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150
head -16 $O/kernel_trace_stats_code2.txt | cut -c1-150

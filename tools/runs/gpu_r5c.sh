#!/bin/bash
# round 4, call 3: Performer block tests (fixture + dropout-ON), code2 bench A/B, kernel traces of pcqm4m and code2
set -u
O=gpurun_out/r5c; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_layer.py -x -q -k "performer or Performer or code2 or golden or fixture" > $O/pytest_perf.log 2>&1; echo "pytest performer rc=$?"; tail -6 $O/pytest_perf.log
timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?"
GPS_FUSED_BLOCK=0 timeout 600 python bench.py --workload code2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $O/bench_code2_op.json 2> $O/bench_code2_op.err; echo "bench code2 operator path rc=$?"
python - <<'PY'
import json
for n in ('bench_code2','bench_code2_op'):
    try:
        d=json.loads(open(f'gpurun_out/r5c/{n}.json').read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],3), d.get('launch_mode'), d.get('launch_trial_ms'))
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
head -45 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-200

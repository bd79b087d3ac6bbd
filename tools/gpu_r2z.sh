#!/bin/bash
set -u
O=gpurun_out/r2v; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_p
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $GRAFT_REPO_ROOT/$O/prof_pcqm4m.json 2> $GRAFT_REPO_ROOT/$O/prof_pcqm4m.log
DB=$(find /tmp/prof_p -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB --top 70 > $O/kernel_trace_stats_pcqm4m.txt 2>&1; fi
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2v/bench_default.json'))
print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d['launch_trial_ms'])
print(json.dumps(d['roofline']))
for kn,v in d['kernels'].items():
    print(kn, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('in_step_ms','isolated_hot_ms','isolated_rotating_ms','frac')})
PY
grep -n "gatedgcn" $O/kernel_trace_stats_pcqm4m.txt | cut -c1-110

#!/bin/bash
# Round-2 evidence run: full GPU test suite, smoke, the default bench line, the code2 line, a clean kernel trace of the
# bench command and the PMC passes.  Outputs under gpurun_out/r2v (copied into profiles/ by hand).
set -u
O=gpurun_out/r2v; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1; fi
  rm -rf /tmp/prof_$w
done
cd $R
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring.txt 2>&1
timeout 200 python tools/gemm_trace.py > $O/gemm_timeline.txt 2>&1
timeout 900 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
python - <<'PY'
import json
for n in ('bench_default','bench_code2','prof_pcqm4m','prof_code2'):
    try:
        d=json.load(open(f'gpurun_out/r2v/{n}.json'))
        print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], json.dumps(d.get('roofline'))[:300])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/rc.txt

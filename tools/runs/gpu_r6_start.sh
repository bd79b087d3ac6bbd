#!/bin/bash
# First call of the NEXT round (written at the end of round 4, not run): re-baseline the tree in one go -- the whole GPU suite
# serially with durations (round 4 ended at ~2.5 minutes of test time), smoke, the three bench lines, kernel traces of the
# pcqm4m / code2 bench commands, the four PMC passes, the loader's host-time profile, and the one A/B round 4 left open:
# GPS_MULTIHOT_WGRAD=1 on the eager step over never-repeating shapes (3.8 ms of rocBLAS host time at
# every first sight of a K = rows GEMM; the switch was only ever measured on a fixed shape).  ~12 minutes of box time.
set -u
O=gpurun_out/r6a; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
GPS_MULTIHOT_WGRAD=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline --steps 20 --warmup 5 > $O/bench_multihot.json 2> $O/bench_multihot.err; echo "bench multihot rc=$?" >> $O/rc.txt
timeout 300 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?" >> $O/rc.txt
timeout 200 python bench.py --workload zinc --no-cpu-baseline > $O/bench_zinc.json 2> $O/bench_zinc.err; echo "bench zinc rc=$?" >> $O/rc.txt
timeout 120 python tools/loader_profile.py 12 > $O/loader_profile.txt 2>&1
echo "t_bench=$(( $(date +%s) - T0 ))" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
    [ $w = pcqm4m ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
timeout 900 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
python - <<'PY'
import json
for n in ('bench_default','bench_multihot','bench_code2','bench_zinc','prof_pcqm4m','prof_code2'):
    try:
        d=json.loads(open(f'gpurun_out/r6a/{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('launch_trial_ms'), json.dumps(d.get('pcie_inclusive_bucketed'))[:300])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/rc.txt; tail -4 $O/loader_profile.txt

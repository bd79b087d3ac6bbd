#!/bin/bash
# Round 5, call 13 (evidence run on the FINAL tree): layer-norm fixtures + whole suite, smoke, the default bench line
# (launch rule, secondary block, roofline, cpu_baseline), core-fork A/B (bwd / both / none), kernel traces of pcqm4m and
# code2, the four PMC passes.
set -u
O=gpurun_out/r6m; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
grep -n "launch-mode trial\|secondary\|timed region\|re-check" $O/bench_default.err
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.4f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run fork_bwd A=1
run fork_both GPS_CORE_FORK=1
run fork_none GPS_CORE_FORK=0
run fork_bwd2 A=1
echo "t_bench=$(( $(date +%s) - T0 ))" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning --no-secondary > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
    [ $w = pcqm4m ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
echo "t_trace=$(( $(date +%s) - T0 ))" >> $O/rc.txt
timeout 900 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt
head -14 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150

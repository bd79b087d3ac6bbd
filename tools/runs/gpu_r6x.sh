#!/bin/bash
# Round 5, evidence run on the FINAL tree: whole GPU suite, smoke, the default bench line (launch rule, secondary block,
# roofline, cpu_baseline, loader legs), kernel traces of pcqm4m and code2.
set -u
O=gpurun_out/r6x; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
grep -n "launch-mode trial\|secondary\|timed region\|re-check\|bucketed loader leg" $O/bench_default.err
echo "t_bench=$(( $(date +%s) - T0 ))" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning --no-secondary > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt
head -12 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150
head -12 $O/kernel_trace_stats_code2.txt | cut -c1-150

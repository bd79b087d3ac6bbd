#!/bin/bash
# Round 5, call 1: whole GPU suite (new: sliced pooling, 100x-shrink replay, reversed-graph yardstick), smoke, the default
# bench line under the new launch rule with the launch probe (eager with the capture alive / destroyed / cache emptied) and
# the `secondary` block, kernel traces of pcqm4m + code2, the four PMC passes (fp16-form ring GEMMs + weight gradients).
set -u
O=gpurun_out/r6a; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
grep -n "replayed vs eager\|un-padded, graphs reversed\|padded vs un-padded\|max|d ff_linear1" $O/pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
GPS_BENCH_LAUNCH_PROBE=1 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
echo "t_bench=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "launch-mode trial\|launch probe\|secondary\|timed region\|re-check" $O/bench_default.err
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
head -12 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150

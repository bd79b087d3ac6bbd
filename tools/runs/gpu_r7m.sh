#!/bin/bash
# Round 5, evidence on the FINAL tree: whole GPU suite, smoke, the default bench line, FAVOR+ probe, kernel traces (pcqm4m, code2)
set -u
O=gpurun_out/r7m; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -n "launch-mode trial\|secondary\|timed region\|re-check\|bucketed loader leg" $O/bench_default.err
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/fv_def
FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_def -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_default.log 2>&1
DB=$(find /tmp/fv_def -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor\|total" | cut -c1-120 > $R/$O/favor_stats_default.txt
cat $R/$O/favor_stats_default.txt
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
head -4 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150

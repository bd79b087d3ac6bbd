#!/bin/bash
# Round 5: the bench + trace half of gpu_r6x.sh again (that call landed on a box whose matrix-pipe kernels ran at half speed)
set -u
O=gpurun_out/r6y; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showclocks --showpower --showperflevel > $O/smi_before.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -n "launch-mode trial\|secondary\|timed region\|re-check\|bucketed loader leg" $O/bench_default.err
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
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
head -8 $O/kernel_trace_stats_pcqm4m.txt | cut -c1-150
head -8 $O/kernel_trace_stats_code2.txt | cut -c1-150
grep -i "sclk\|mclk\|power\|perf" $O/smi_before.txt | head -8

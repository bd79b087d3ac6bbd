#!/bin/bash
# round 6: kernel trace of the pcqm4m step (90 steps): per-kernel table + one step's timeline.   usage: r6_prof.sh OUTDIR [code2]
set -u
OUT=$1; mkdir -p $OUT
export TMPDIR=/tmp
Q="--steps 90 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
rocprofv3 --kernel-trace -d /tmp/prof_r6 -o bench -- python bench.py $Q > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
DB=$(find /tmp/prof_r6 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB --top 70 > $OUT/kernel_trace_stats_pcqm4m.txt
python tools/rocpd_timeline.py $DB --full > $OUT/timeline_pcqm4m.txt 2>&1
head -45 $OUT/kernel_trace_stats_pcqm4m.txt | cut -c1-190

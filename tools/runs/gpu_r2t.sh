#!/bin/bash
set -u
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_s0
GPS_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s0 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-kernel-roofline > $GRAFT_REPO_ROOT/$O/bench_side0.json 2> $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
find /tmp/prof_s0 -name "*kernel_stats*" | head
F=$(find /tmp/prof_s0 -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp $F $O/kernel_stats_side0.csv; fi
DB=$(find /tmp/prof_s0 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/stats_side0.txt 2>&1; fi
head -40 $O/stats_side0.txt
tail -3 $O/prof.log

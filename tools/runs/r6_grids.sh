#!/bin/bash
# launch shapes / dispatch rounds of every kernel of the pcqm4m and code2 steps (tools/rocpd_grids.py over a short kernel trace)
set -u
OUT=gpurun_out/${1:-r6_grids}; mkdir -p $OUT
export TMPDIR=/tmp
for wl in pcqm4m code2; do
  Q="--workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
  rm -rf /tmp/prof_g; rocprofv3 --kernel-trace -d /tmp/prof_g -o bench -- python bench.py $Q > $OUT/$wl.json 2> $OUT/$wl.err
  python tools/rocpd_grids.py $(find /tmp/prof_g -name "*.db" | head -1) --top 45 --objs graphgps_amd/csrc/*.o > $OUT/grids_$wl.txt 2>&1
done
head -30 $OUT/grids_pcqm4m.txt | cut -c1-170

#!/bin/bash
# Round 5: FAVOR+ forms at the dataset's own graph sizes (CODE2_REAL: ~125 nodes per graph), 32 and 128 graphs per batch:
# context kernels per-wavefront vs staged -- does the default (staged from 64 rows per graph on) pick the faster one?
set -u
O=gpurun_out/r7o; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp; cd /tmp
for nb in 32 128; do
for mode in plainctx staged; do
  rm -rf /tmp/fv_x
  C=1; [ $mode = plainctx ] && C=0
  env FAVOR_PROFILE=CODE2_REAL FAVOR_GRAPHS=$nb GPS_FAVOR_CTX_LDS=$C FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_x -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_${nb}_$mode.log 2>&1
  DB=$(find /tmp/fv_x -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor\|total" | cut -c1-110 > $R/$O/favor_stats_${nb}_$mode.txt
  echo "== $nb graphs, $mode"; cat $R/$O/favor_stats_${nb}_$mode.txt
done; done

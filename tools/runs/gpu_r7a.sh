#!/bin/bash
# Round 5: FAVOR+ query-side kernel with the context record staged in LDS too (chunks of four tiles per graph and head)
set -u
O=gpurun_out/r7a; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check rc=$?"; cat $O/check.txt; tail -3 $O/check.err
export TMPDIR=/tmp; cd /tmp
for mode in lds lc; do
  rm -rf /tmp/fv_$mode
  L=0; [ $mode = lc ] && L=1
  GPS_FAVOR_LDS=1 GPS_FAVOR_LC=$L FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor\|total" | cut -c1-120 > $R/$O/favor_stats_$mode.txt
  echo "== mode $mode"; cat $R/$O/favor_stats_$mode.txt
done

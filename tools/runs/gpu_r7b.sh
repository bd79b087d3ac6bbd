#!/bin/bash
# Round 5: FAVOR+ query side, context record in LDS: with / without the per-tile scheduling fences
set -u
O=gpurun_out/r7b; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPS_FAVOR_LC_FENCE=0 timeout 300 python tools/favor_lds_check.py > $O/check_nofence.txt 2> $O/check.err; echo "check(no fence) rc=$?"; grep -v "vs plain" $O/check_nofence.txt
export TMPDIR=/tmp; cd /tmp
for mode in fence nofence; do
  rm -rf /tmp/fv_$mode
  F=1; [ $mode = nofence ] && F=0
  GPS_FAVOR_LDS=1 GPS_FAVOR_LC=1 GPS_FAVOR_LC_FENCE=$F FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor_bwd_q\|total" | cut -c1-120 > $R/$O/favor_stats_$mode.txt
  echo "== mode $mode"; cat $R/$O/favor_stats_$mode.txt
done

#!/bin/bash
# Round 5: FAVOR+ query side: LP / LC 4 waves / LC 4 waves + next record prefetched through registers / LC 8 waves, one box
set -u
O=gpurun_out/r7d; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check_pre.txt 2> $O/check.err; echo "check(prefetch form = default LC) rc=$?"; grep -v "vs plain" $O/check_pre.txt
export TMPDIR=/tmp; cd /tmp
for mode in lp lc4 lc4pre lc8 lp2 lc4pre2; do
  rm -rf /tmp/fv_$mode
  case $mode in
    lp*) E="GPS_FAVOR_LC=0";;
    lc4pre*) E="GPS_FAVOR_LC=1 GPS_FAVOR_LC_PREFETCH=1";;
    lc4) E="GPS_FAVOR_LC=1 GPS_FAVOR_LC_PREFETCH=0";;
    lc8) E="GPS_FAVOR_LC=1 GPS_FAVOR_LC_WAVES=8";;
  esac
  env GPS_FAVOR_LDS=1 $E FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor_bwd_q" | cut -c1-120 > $R/$O/favor_stats_$mode.txt
  echo "== $mode"; cat $R/$O/favor_stats_$mode.txt
done

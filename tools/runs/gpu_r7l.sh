#!/bin/bash
# Round 5: staged FAVOR+ context kernels: 9 / 12 / 16 wavefronts per workgroup; identity at the default (12)
set -u
O=gpurun_out/r7l; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check(12) rc=$?"; grep -v "vs plain" $O/check.txt | cut -c1-120
GPS_FAVOR_CTX_WAVES=16 timeout 300 python tools/favor_lds_check.py > $O/check16.txt 2>> $O/check.err; echo "check(16) rc=$?"
export TMPDIR=/tmp; cd /tmp
for mode in 9 12 16 12b; do
  rm -rf /tmp/fv_$mode
  env GPS_FAVOR_CTX_LDS=1 GPS_FAVOR_CTX_WAVES=${mode%b} FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "ctx\|total" | cut -c1-110 > $R/$O/favor_stats_$mode.txt
  echo "== waves $mode"; cat $R/$O/favor_stats_$mode.txt
done

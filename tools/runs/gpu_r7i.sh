#!/bin/bash
# Round 5: FAVOR+ context kernels: wavefront -> work order (feature tile fastest vs slice fastest) x wavefronts per workgroup
set -u
O=gpurun_out/r7i; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check rc=$?"; grep -v "vs plain" $O/check.txt | head -4
export TMPDIR=/tmp; cd /tmp
for mode in old_4 new_4 new_8 new_16 old_16 new_4b; do
  rm -rf /tmp/fv_$mode
  T=1; [ ${mode%%_*} = old ] && T=0
  W=${mode##*_}; W=${W%b}
  env GPS_FAVOR_CTX_TILE_FIRST=$T GPS_FAVOR_CTX_WAVES=$W FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "ctx\|total" | cut -c1-110 > $R/$O/favor_stats_$mode.txt
  echo "== $mode"; cat $R/$O/favor_stats_$mode.txt
done

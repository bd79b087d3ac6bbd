#!/bin/bash
# Round 5: FAVOR+ context kernels with their rows staged in LDS (a workgroup per (graph, head, slice)): identity, times
set -u
O=gpurun_out/r7k; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check rc=$?"; cat $O/check.txt | cut -c1-170; tail -3 $O/check.err
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "favor" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
export TMPDIR=/tmp; cd /tmp
for mode in plainctx staged plainctx2 staged2; do
  rm -rf /tmp/fv_$mode
  C=1; [ ${mode%%[0-9]*} = plainctx ] && C=0
  env GPS_FAVOR_CTX_LDS=$C FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "ctx\|sum_parts\|total" | cut -c1-110 > $R/$O/favor_stats_$mode.txt
  echo "== $mode"; cat $R/$O/favor_stats_$mode.txt
done

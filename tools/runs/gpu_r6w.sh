#!/bin/bash
# Round 5: FAVOR+ LDS projection: bit-identity after the explicit fma; query-side kernel at 4 vs 8 wavefronts per CU
set -u
O=gpurun_out/r6w; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check rc=$?"; grep -v "mode" $O/check.txt; tail -3 $O/check.err
GPS_FAVOR_Q_THREADS=512 timeout 300 python tools/favor_lds_check.py > $O/check512.txt 2> $O/check512.err; echo "check512 rc=$?"; grep -v "mode" $O/check512.txt
export TMPDIR=/tmp; cd /tmp
for mode in 0 1 512; do
  rm -rf /tmp/fv_$mode
  L=1; [ $mode = 0 ] && L=0
  Q=256; [ $mode = 512 ] && Q=512
  GPS_FAVOR_LDS=$L GPS_FAVOR_Q_THREADS=$Q FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor\|total" | cut -c1-120 > $R/$O/favor_stats_$mode.txt
  echo "== mode $mode"; cat $R/$O/favor_stats_$mode.txt
done
cd $R
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -x -q -m gpu -k "favor or performer or Performer" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
GPS_FAVOR_LDS=1 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -x -q -m gpu -k "favor or performer or Performer" > $O/tests_lds1.log 2>&1; echo "tests(lds forced) rc=$?"; grep -n "passed\|failed" $O/tests_lds1.log | tail -2

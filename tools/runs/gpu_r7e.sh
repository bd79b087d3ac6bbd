#!/bin/bash
# Round 5: FAVOR+ output / query-side / key-side kernels with the (graph, head) record staged in LDS: identity + per-kernel times
set -u
O=gpurun_out/r7e; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/favor_lds_check.py > $O/check.txt 2> $O/check.err; echo "check(default LC) rc=$?"; grep -v "vs plain" $O/check.txt; tail -2 $O/check.err
GPS_FAVOR_LC_WAVES=8 timeout 300 python tools/favor_lds_check.py > $O/check_w8.txt 2>> $O/check.err; echo "check(8 waves) rc=$?"; grep -v "vs plain" $O/check_w8.txt | head -2
export TMPDIR=/tmp; cd /tmp
for mode in lp pre4 w8 pre4b; do
  rm -rf /tmp/fv_$mode
  case $mode in
    lp) E="GPS_FAVOR_LC=0";;
    pre4*) E="GPS_FAVOR_LC=1";;
    w8) E="GPS_FAVOR_LC=1 GPS_FAVOR_LC_WAVES=8";;
  esac
  env GPS_FAVOR_LDS=1 $E FAVOR_ITERS=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$mode -o fv -- python $R/tools/favor_probe.py > $R/$O/probe_$mode.log 2>&1
  DB=$(find /tmp/fv_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB --top 20 2>&1 | grep -i "favor\|total" | cut -c1-120 > $R/$O/favor_stats_$mode.txt
  echo "== $mode"; cat $R/$O/favor_stats_$mode.txt
done
cd $R
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py tests/test_hip_padding.py -x -q -m gpu -k "favor or performer or Performer or code2" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.log | tail -2
GPS_FAVOR_LC=1 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py tests/test_hip_padding.py -x -q -m gpu -k "favor or performer or Performer or code2" > $O/tests_lc1.log 2>&1; echo "tests(LC forced) rc=$?"; grep -n "passed\|failed" $O/tests_lc1.log | tail -2

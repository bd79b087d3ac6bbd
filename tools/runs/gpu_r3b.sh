#!/bin/bash
# round 3: the failing norm test again + a per-dispatch timeline of the step (default and all-switches-off)
set -u
O=gpurun_out/r3b; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_hip_norm.py -m gpu -q -p no:cacheprovider > $O/pytest_norm.log 2>&1; echo "norm rc=$?" > $O/rc.txt
tail -3 $O/pytest_norm.log
export TMPDIR=/tmp
cd /tmp
for cfg in "" "GPS_GG_STATS=0 GPS_GEMM_STATS=0"; do
  tag=$(echo "$cfg" | tr ' =' '__'); tag=${tag:-default}
  rm -rf /tmp/prof_$tag
  env $cfg timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_$tag.json 2> $R/$O/prof_$tag.log
  DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$tag.txt 2>&1
    python $R/tools/rocpd_stats.py $DB --top 40 > $R/$O/stats_$tag.txt 2>&1
  fi
  rm -rf /tmp/prof_$tag
done
cd $R
cat $O/rc.txt
grep "^#" $O/timeline_default.txt | head -5

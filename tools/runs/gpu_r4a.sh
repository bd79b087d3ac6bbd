#!/bin/bash
# round 3: what sits in the idle gaps of the replayed step (kernel + memory-copy trace); 304-wide layer test with attribution
set -u
O=gpurun_out/r4a; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "baseline_sizes" -s > $O/pytest.log 2>&1; echo "tests rc=$?"
grep -n "kink\|second pass\|passed\|failed" $O/pytest.log | tail -12
export TMPDIR=/tmp
cd /tmp
for mode in graph eager; do
  rm -rf /tmp/prof_$mode
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_$mode -o bench -- python $R/bench.py --steps 12 --warmup 4 --launch $mode --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $R/$O/prof_$mode.json 2> $R/$O/prof_$mode.log
  DB=$(find /tmp/prof_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$mode.txt 2>&1
  grep -n "^# step\|^# some" $R/$O/timeline_$mode.txt
  grep -n "memcpy" $R/$O/timeline_$mode.txt | head -20
done
cd $R
for mode in graph eager graph eager; do
  timeout 300 python bench.py --steps 30 --warmup 10 --launch $mode --no-cpu-baseline --no-kernel-roofline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['ms_per_step'],3), d.get('host_enqueue_ms_per_step'))"
done

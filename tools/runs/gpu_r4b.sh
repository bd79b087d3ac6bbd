#!/bin/bash
# round 3: two alternating instances of the captured step (host launch hidden behind the previous replay) vs one; eager
set -u
O=gpurun_out/r4b; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_layer.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -k "baseline_sizes or capture or replay or graph or trainstep" -s > $O/pytest.log 2>&1; echo "tests rc=$?"
grep -n "second pass\|passed\|failed" $O/pytest.log | tail -8
for i in 1 2; do
for cfg in "GPS_CAPTURE_COPIES=2 --launch graph" "GPS_CAPTURE_COPIES=1 --launch graph" "GPS_CAPTURE_COPIES=3 --launch graph" "X=1 --launch eager"; do
  set -- $cfg
  env $1 timeout 300 python bench.py --steps 30 --warmup 10 $2 $3 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg 2>$O/err_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],3), d.get('host_enqueue_ms_per_step'), d['launch_mode'])"
done
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_g2
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g2 -o bench -- python $R/bench.py --steps 12 --warmup 4 --launch graph --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $R/$O/prof_graph2.json 2> $R/$O/prof_graph2.log
DB=$(find /tmp/prof_g2 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_graph2.txt 2>&1
grep -n "^# step\|^# some" $R/$O/timeline_graph2.txt
sed -n 16,22p $R/$O/timeline_graph2.txt | cut -c1-100

#!/bin/bash
# round 3: norm tests + layer tests, bench A/B (row blocks per task, weight gradients on a side stream), one timeline
set -u
O=gpurun_out/r3c; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_hip_norm.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
for cfg in "" "GPS_NORM_BLOCKS=512" "GPS_BLOCK_WGRAD_SIDE_STREAM=1" "GPS_BLOCK_WGRAD_SIDE_STREAM=1 GPS_WGRAD_TARGET_BLOCKS=256"; do
  tag=$(echo "$cfg" | tr ' =' '__'); tag=${tag:-default}
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "== $tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('launch_trial_ms'), d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_d
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_d -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_default.json 2> $R/$O/prof_default.log
DB=$(find /tmp/prof_d -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_default.txt 2>&1
cd $R
cat $O/rc.txt

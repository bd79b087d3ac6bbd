#!/bin/bash
# round 3, first GPU call: the new in-launch reductions (norm lists, GatedGCN / GEMM statistics), then the whole suite,
# then the bench line with its A/B switches.
set -u
O=gpurun_out/r3a; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_norm.py -m gpu -q -x -p no:cacheprovider > $O/pytest_norm.log 2>&1; echo "norm rc=$?" > $O/rc.txt
tail -5 $O/pytest_norm.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_hip_norm.py > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/rc.txt
tail -5 $O/pytest_gpu.log
for cfg in "" "GPS_GG_STATS=0" "GPS_GEMM_STATS=0" "GPS_GG_STATS=0 GPS_GEMM_STATS=0"; do
  tag=$(echo "$cfg" | tr ' =' '__'); tag=${tag:-default}
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "== $tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('launch_trial_ms'), d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"
done
cat $O/rc.txt

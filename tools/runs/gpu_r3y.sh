#!/bin/bash
# round 3: EDGE variants of the ring GEMM (d = 304 / 96 / 48); fresh kernel timeline of the default (single-stream) step
set -u
O=gpurun_out/r3y; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py tests/test_hip_norm.py -m gpu -q -p no:cacheprovider -k "gemm or baseline_sizes or dropout_on or race or fixture" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -5 $O/pytest.log
for gb in "304,7569,15348" "96,7569,15348"; do
  GEMM_BENCH=$gb timeout 300 python tools/gemm_panel_bench.py > $O/gemm_$(echo $gb | cut -d, -f1).txt 2>&1
  tail -16 $O/gemm_$(echo $gb | cut -d, -f1).txt
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_d
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg > $R/$O/prof_default.json 2> $R/$O/prof_default.log
DB=$(find /tmp/prof_d -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_default.txt 2>&1
find /tmp/prof_d -name "*stats*" | head; cp $(find /tmp/prof_d -name "*kernel_stats*" | head -1) $R/$O/ 2>/dev/null
cd $R
grep -n "per kernel" -A28 $O/timeline_default.txt | head -40

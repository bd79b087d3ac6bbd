#!/bin/bash
# Round 5: the default bench line of the final tree once more (box-to-box spread of the last lines: 8.38 / 8.83 ms)
set -u
O=gpurun_out/r7n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -n "launch-mode trial\|secondary\|timed region\|re-check\|bucketed loader leg" $O/bench_default.err

#!/bin/bash
# Round 5, last call: the whole GPU suite + smoke + the default bench line on the tree as committed.
set -u
O=gpurun_out/r6o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_gpu.log; grep -n "^FAILED" $O/pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -n "launch-mode trial\|secondary\|timed region" $O/bench_default.err

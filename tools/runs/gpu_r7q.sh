#!/bin/bash
# Round 5: the whole GPU suite + smoke on the last commit
set -u
O=gpurun_out/r7q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2; grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"

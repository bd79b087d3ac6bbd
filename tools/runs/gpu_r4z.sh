#!/bin/bash
# final tree: whole GPU suite + smoke
set -u
O=gpurun_out/r4z; mkdir -p $O
timeout 340 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

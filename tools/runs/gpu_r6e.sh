#!/bin/bash
# Round 5, call 5: padded batches through the GINE and Performer blocks (new tests), then the whole suite on the current tree.
set -u
O=gpurun_out/r6e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 500 python -m pytest tests/test_hip_padding.py -q -p no:cacheprovider -s > $O/pytest_padding.log 2>&1; echo "pytest padding rc=$?"
grep -n "passed\|failed\|FAILED\|Error\|replayed\|padded vs un-padded\|graphs reversed" $O/pytest_padding.log | head -30
echo "t_pad=$(( $(date +%s) - T0 ))"
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_hip_padding.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -3 $O/pytest_all.log; grep -n "^FAILED" $O/pytest_all.log | head
echo "t_all=$(( $(date +%s) - T0 ))"

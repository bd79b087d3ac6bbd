#!/bin/bash
# Round 5, call 7: evaluation replay (EvalStep) + layer-norm fixtures, then the whole suite on the final tree.
set -u
O=gpurun_out/r6g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 400 python -m pytest tests/test_hip_padding.py tests/test_hip_layer.py -q -p no:cacheprovider -s -k "eval_epoch or layernorm or eval_mode" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"
grep -n "passed\|failed\|FAILED\|Error\|replayed" $O/pytest_new.log | head -20
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -3 $O/pytest_all.log; grep -n "^FAILED" $O/pytest_all.log | head
echo "t_all=$(( $(date +%s) - T0 ))"

#!/bin/bash
# Round 4, last call (the ~3.7 GPU-minutes left): the test files not re-run since the round's changes -- test_hip_ops.py,
# the GPU forms of test_oracle_golden.py, and test_hip_layer.py without the four tests the third / fourth calls covered.
set -u
O=gpurun_out/r5k; mkdir -p $O
T0=$(date +%s)
timeout 212 python -m pytest tests/test_hip_ops.py tests/test_oracle_golden.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider --durations=15 \
  --deselect "tests/test_hip_layer.py::test_full_model_with_dropout_on_vs_masked_oracle" \
  --deselect "tests/test_hip_layer.py::test_full_model_vs_oracle" \
  --deselect "tests/test_hip_layer.py::test_code2_model_vs_oracle" > $O/pytest_rest.log 2>&1; echo "pytest_rest rc=$?" > $O/rc.txt
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_rest.log | tail -2
grep -n "^FAILED\|^ERROR\|^E  " $O/pytest_rest.log | head -30
grep -n "s call " $O/pytest_rest.log | head -16
tail -3 $O/pytest_rest.log
cat $O/rc.txt

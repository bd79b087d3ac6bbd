#!/bin/bash
# Round 4, third call (4 GPU-minutes): the one failure of the second call's full run (#26), the slowest test, the op-level
# padded tests with the salt fixture, and the host-time profile of the bucketed loader leg.
set -u
O=gpurun_out/r5i; mkdir -p $O
T0=$(date +%s)
timeout 210 python -m pytest "tests/test_hip_layer.py::test_full_model_vs_oracle" \
  "tests/test_hip_layer.py::test_full_model_with_dropout_on_vs_masked_oracle" \
  "tests/test_hip_layer.py::test_gpslayer_vs_oracle_baseline_sizes[CustomGatedGCN-Transformer-384-16-P30-256]" \
  "tests/test_hip_layer.py::test_code2_model_vs_oracle" \
  tests/test_hip_padding.py "tests/test_hip_norm.py::test_norm_lists_on_padded_batches_see_the_real_rows_only" \
  "tests/test_hip_norm.py::test_gemm_statistics_epilogue_skips_padding_rows" \
  -m gpu -q -s -p no:cacheprovider --durations=0 > $O/pytest_sel.log 2>&1; echo "pytest_sel rc=$?" > $O/rc.txt
echo "t_sel=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_sel.log | tail -2
grep -n "^FAILED\|^ERROR\|^E  " $O/pytest_sel.log | head -30
grep -n "s call " $O/pytest_sel.log | head -30
timeout 100 python tools/loader_profile.py 12 > $O/loader_profile.txt 2>&1; echo "profile rc=$?" >> $O/rc.txt
cat $O/loader_profile.txt | tail -5
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
cat $O/rc.txt

#!/bin/bash
# round 3: ring GEMM for N, K multiples of 4 (d = 52 / 72)
set -u
O=gpurun_out/r4f; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gemm_panel_fp32 and (500-52-364 or 500-364-52 or 743-72-72 or 300-52-52 or 500-20-36 or 2000-304-304 or 1000-384-384 or 130-16-16 or 743-96-96)" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/pytest_ops.log
timeout 300 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "baseline_sizes and (52 or 96-4)" > $O/pytest_layer.log 2>&1; echo "layer rc=$?"; tail -2 $O/pytest_layer.log

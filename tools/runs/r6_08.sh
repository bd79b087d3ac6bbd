#!/bin/bash
# round 6: paired GEMM dispatch (forward C(e) | merged projection, backward g_pq Wcat | g_ce W_C): parity + A/B
set -u
OUT=gpurun_out/r6_08; mkdir -p $OUT
python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "gemm_panel_pair or (gemm_panel_split and 7569)" > $OUT/pair.log 2>&1; echo "pair rc=$?"; tail -3 $OUT/pair.log
python -m pytest tests/test_hip_layer.py tests/test_hip_padding.py -q -x -m gpu > $OUT/layer.log 2>&1; echo "layer rc=$?"; tail -3 $OUT/layer.log
bash tools/runs/r6_ab.sh $OUT "pair:" "nopair:GPS_GEMM_PAIR=0"

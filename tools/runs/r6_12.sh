#!/bin/bash
# round 6 experiment: ring GEMM with a 3-slot ring (all shapes) and 64 x 128 tiles at two workgroups per CU
set -u
OUT=gpurun_out/r6_12; mkdir -p $OUT
X=$PWD/tools/micro/exp/libgps_hip_exp.so
echo "== default library" | tee $OUT/gemm.txt
python tools/gemm_panel_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm.txt
echo "== 3-slot ring, default tiles" | tee -a $OUT/gemm.txt
GPS_HIP_LIB=$X python tools/gemm_panel_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm.txt
echo "== 3-slot ring, 64 x 128 tiles (two workgroups per CU)" | tee -a $OUT/gemm.txt
GPS_HIP_LIB=$X GPS_EXP_NJ=2 GPS_EXP_MB=1 python tools/gemm_panel_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm.txt
echo "== 3-slot ring, 128 x 128 tiles" | tee -a $OUT/gemm.txt
GPS_HIP_LIB=$X GPS_EXP_NJ=2 GPS_EXP_MB=2 python tools/gemm_panel_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm.txt

#!/bin/bash
# Ablation builds of the fp16-form ring GEMM (csrc/gemm_panel.hip, k_gemm_ring16): the same library with parts of the main
# loop compiled out (-DGPS_ABL_R16_*: results are garbage, timing only), one libgps_hip.so per variant under
# tools/micro/abl_ring/ (git-ignored; they travel to the GPU box with gpurun).  Run HERE (no GPU needed to build), then
# on the box:  for v in tools/micro/abl_ring/*.so; do GPS_HIP_LIB=$v python tools/ring_ablate_bench.py; done
set -u
cd "$(dirname "$0")/../../graphgps_amd/csrc"
OUT=../../tools/micro/abl_ring; mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function"
OTHERS=$(ls *.o | grep -v '^gemm_panel.o$')
VARIANTS=${VARIANTS:-"base: nodma:-DGPS_ABL_R16_NO_DMA nosplit:-DGPS_ABL_R16_NO_SPLIT nord:-DGPS_ABL_R16_NO_RD nobar:-DGPS_ABL_R16_NO_BAR nostore:-DGPS_ABL_R16_NO_STORE nomfma:-DGPS_ABL_R16_NO_MFMA nodma_nosplit:-DGPS_ABL_R16_NO_DMA,-DGPS_ABL_R16_NO_SPLIT onlymfma:-DGPS_ABL_R16_NO_DMA,-DGPS_ABL_R16_NO_SPLIT,-DGPS_ABL_R16_NO_RD,-DGPS_ABL_R16_NO_BAR"}
pids=""
for v in $VARIANTS; do
  name=${v%%:*}; defs=$(echo "${v#*:}" | tr ',' ' ')
  ( $HIPCC $FLAGS $defs -c gemm_panel.hip -o $OUT/gemm_panel_$name.o && \
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/$name.so $OTHERS $OUT/gemm_panel_$name.o && echo "built $name" ) &
  pids="$pids $!"
  while [ $(jobs -r | wc -l) -ge ${JOBS:-4} ]; do sleep 2; done
done
wait
ls -la $OUT/*.so

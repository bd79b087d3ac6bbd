#!/bin/bash
# Build the ablation variants of csrc/gemm_split.hip (run here, no GPU needed) -> tools/micro/abl_*;
# `tools/micro/gemm_ablate.sh run [R K M]` executes them (on the GPU box).
set -e
cd "$(dirname "$0")"
SRC=../../graphgps_amd/csrc
VARIANTS="base: nosplit:-DGPS_ABL_NO_SPLIT nomfma:-DGPS_ABL_NO_MFMA noldsread:-DGPS_ABL_NO_LDSREAD noglobal:-DGPS_ABL_NO_GLOBAL onlymfma:-DGPS_ABL_NO_LDSREAD,-DGPS_ABL_NO_GLOBAL,-DGPS_ABL_NO_SPLIT mfma_bar:-DGPS_ABL_NO_LDSREAD,-DGPS_ABL_NO_GLOBAL,-DGPS_ABL_NO_SPLIT,-DGPS_ABL_NO_LDSWRITE mfma_nobar:-DGPS_ABL_NO_LDSREAD,-DGPS_ABL_NO_GLOBAL,-DGPS_ABL_NO_SPLIT,-DGPS_ABL_NO_LDSWRITE,-DGPS_ABL_NO_BARRIER nowrite:-DGPS_ABL_NO_LDSWRITE"
if [ "$1" = "run" ]; then
  shift
  for v in $VARIANTS; do n=${v%%:*}; ./abl_$n "${1:-7569}" "${2:-384}" "${3:-2688}" $n; done
  exit 0
fi
for v in $VARIANTS; do
  n=${v%%:*}; f=$(echo "${v#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPS_ABLATION_BUILD -I../../include -I$SRC $f \
      gemm_ablate.hip $SRC/gemm_split.hip $SRC/gps_common.hip -o abl_$n &
done
wait
ls -la abl_*

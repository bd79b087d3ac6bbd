#!/bin/bash
# rocprofv3 --pmc passes over tools/kernel_probe.py (one counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit
# one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"); per-kernel means -> $1/pmc_<group>.txt
set -u
OUT=${1:-gpurun_out/pmc}; mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o probe -- python $ROOT/tools/kernel_probe.py > $ROOT/$OUT/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $ROOT/tools/rocpd_pmc.py $DB --match "k_gatedgcn|k_sattn|k_attn|k_wgrad|k_gemm_panel|k_gemm_ring|k_absmax" > $ROOT/$OUT/pmc_$i.txt 2>&1; fi
  rm -rf /tmp/pmc_$i
done
cd $ROOT
python tools/pmc_summarize.py $OUT > $OUT/summary.json 2> $OUT/summary.err
head -c 3000 $OUT/summary.json

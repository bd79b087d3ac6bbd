#!/bin/bash
# Round 5: PMC passes over tools/favor_probe.py (FAVOR+ kernels in their default forms): matrix-pipe busy, issue stalls, LDS bank
# conflicts of the pitch-68 staging (one counter group per pass, kernel-trace only)
set -u
O=gpurun_out/r7j; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  FAVOR_ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o probe -- python $R/tools/favor_probe.py > $R/$O/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB --match "k_favor" > $R/$O/pmc_$i.txt 2>&1; fi
  rm -rf /tmp/pmc_$i
done
cd $R
wc -l $O/pmc_*.txt
grep -h "bwd_q_lc\|out_lc\|bwd_k_lc" $O/pmc_2.txt | cut -c1-150 | head -30

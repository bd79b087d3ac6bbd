#!/bin/bash
# round 3: optimizer / loader / shape-cache tests, code2 model test, default bench (host-batch leg before capture),
# FAVOR+ PMC passes (none existed), timeline of the step
set -u
O=gpurun_out/r3f; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_optim.py -m gpu -q -p no:cacheprovider > $O/pytest_optim.log 2>&1; echo "optim rc=$?" > $O/rc.txt
tail -4 $O/pytest_optim.log
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -s -k "code2_model" > $O/pytest_code2.log 2>&1; echo "code2 rc=$?" >> $O/rc.txt
grep -E "passed|failed|code2 model|per-graph|ReLU sign|Error" $O/pytest_code2.log | tail -8
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
grep -E "host-batch|launch-mode|timed region" $O/bench_default.err
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcf_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcf_$i -o probe -- python $R/tools/favor_probe.py > $R/$O/favor_pmc_$i.log 2>&1
  DB=$(find /tmp/pmcf_$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB --match "k_favor" > $R/$O/favor_pmc_$i.txt 2>&1; fi
  rm -rf /tmp/pmcf_$i
done
cd $R
head -30 $O/favor_pmc_1.txt
cat $O/rc.txt

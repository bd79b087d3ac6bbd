#!/bin/bash
set -u
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider --maxfail=12 -k "attention or gemm_panel" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -12 $O/pytest.log
timeout 600 python tools/gemm_panel_bench.py > $O/gemm_panel.txt 2>&1; echo "gemm rc=$?" >> $O/rc.txt; cat $O/gemm_panel.txt | grep -v Tunable | tail -14
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py > $GRAFT_REPO_ROOT/$O/prof_probe.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kp -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB --top 6 > $O/stats_probe.txt 2>&1; cat $O/stats_probe.txt | cut -c1-150
cat $O/rc.txt

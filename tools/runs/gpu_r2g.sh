#!/bin/bash
set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider --maxfail=12 -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?" > $O/rc.txt
tail -4 $O/pytest_attn.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py > $GRAFT_REPO_ROOT/$O/prof_probe.log 2>&1; cd $GRAFT_REPO_ROOT
tail -2 $O/prof_probe.log
DB=$(find /tmp/kp -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB --top 8 > $O/stats_probe.txt 2>&1; cat $O/stats_probe.txt | cut -c1-160
cat $O/rc.txt

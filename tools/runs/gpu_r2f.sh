#!/bin/bash
set -u
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/kernel_probe.py > $O/probe.txt 2>&1; echo "probe rc=$?" > $O/rc.txt
cat $O/probe.txt | tail -5
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py > $GRAFT_REPO_ROOT/$O/prof_probe.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kp -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB --top 20 > $O/stats_probe.txt 2>&1; cat $O/stats_probe.txt | cut -c1-200
GG_COPY=1 timeout 300 python tools/gg_sweep.py > $O/copy.txt 2>&1; cat $O/copy.txt
timeout 900 python tools/gg_sweep.py > $O/gg_sweep.txt 2>&1; cat $O/gg_sweep.txt
cat $O/rc.txt

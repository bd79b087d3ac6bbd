#!/bin/bash
# round-2 GPU call A: parity suite, headline bench, code2 bench + kernel trace
set -u
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench_pcqm4m.json 2> $O/bench_pcqm4m.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --workload code2 --steps 10 --warmup 3 > $O/bench_code2.json 2> $O/bench_code2.err; echo "code2 rc=$?" >> $O/rc.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_code2 -o code2 -- python $GRAFT_REPO_ROOT/bench.py --workload code2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --launch eager > $GRAFT_REPO_ROOT/$O/prof_code2.log 2>&1; echo "prof rc=$?" >> $GRAFT_REPO_ROOT/$O/rc.txt
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_code2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --top 50 > $O/stats_code2.txt 2>&1
find $O/prof_code2 -name "*.db" -size +20M -delete
ls -R $O/prof_code2 | head -20
cat $O/rc.txt
head -c 1500 $O/bench_pcqm4m.json

#!/bin/bash
# round 6 evidence run on the final tree: suite, smoke, the driver-form bench line, kernel traces (pcqm4m + code2), PMC passes
set -u
OUT=gpurun_out/${1:-r6_ev}; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/ -q -m gpu --durations=15 > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?" | tee $OUT/summary.txt; tail -3 $OUT/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
grep -E "timed region|launch-mode|secondary" $OUT/bench.err | cut -c1-200
bash tools/runs/r6_prof.sh $OUT > $OUT/prof.txt 2>&1; echo "prof rc=$?" | tee -a $OUT/summary.txt
# code2 step trace
Q="--workload code2 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary"
rm -rf /tmp/prof_c2; rocprofv3 --kernel-trace -d /tmp/prof_c2 -o bench -- python bench.py $Q > $OUT/prof_code2_bench.json 2> $OUT/prof_code2.err; echo "prof code2 rc=$?" | tee -a $OUT/summary.txt
python tools/rocpd_stats.py $(find /tmp/prof_c2 -name "*.db" | head -1) --top 50 > $OUT/kernel_trace_stats_code2.txt
bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc.txt 2>&1; echo "pmc rc=$?" | tee -a $OUT/summary.txt
head -30 $OUT/kernel_trace_stats_pcqm4m.txt | cut -c1-160

#!/bin/bash
# Round-3 evidence run: full GPU test suite, smoke, the default bench line, code2 / zinc lines, kernel traces (+ timeline) of
# the bench command, GEMM tables, PMC passes (+ one FETCH_SIZE pass with the round-2 weight-gradient work-item order).
# Outputs under gpurun_out/r3v (copied into profiles/ by hand).
set -u
O=gpurun_out/r3v; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --workload zinc --no-cpu-baseline > $O/bench_zinc.json 2> $O/bench_zinc.err; echo "bench zinc rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
    python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring_384.txt 2>&1
GEMM_BENCH=304,7569,15348 timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring_304.txt 2>&1
GEMM_BENCH=256,25600,76800 timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring_256.txt 2>&1
timeout 900 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
# the weight-gradient kernel's fabric traffic with the round-2 work-item order, for the A/B in DESIGN 4.5c
cd /tmp; rm -rf /tmp/pmc_x
GPS_WGRAD_XCD_MAP=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_x -o probe -- python $R/tools/kernel_probe.py > $R/$O/pmc_xcd0.log 2>&1
DB=$(find /tmp/pmc_x -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB --match "k_wgrad" > $R/$O/pmc_wgrad_xcdmap0.txt 2>&1
cd $R
python - <<'PY'
import json
for n in ('bench_default','bench_code2','bench_zinc','prof_pcqm4m','prof_code2'):
    try:
        d=json.loads(open(f'gpurun_out/r3v/{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('pcie_inclusive_ms_per_step'), json.dumps(d.get('roofline'))[:300])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/rc.txt

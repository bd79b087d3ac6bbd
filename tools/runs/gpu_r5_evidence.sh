#!/bin/bash
# Round-4 evidence run: full GPU test suite, smoke, the default bench line, code2 / zinc lines, kernel traces of the bench
# command, GEMM + weight-gradient tables, PMC passes, the launch-form A/B.  Outputs under gpurun_out/r5v (copied into profiles/).
set -u
O=gpurun_out/r5v; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --workload zinc --no-cpu-baseline > $O/bench_zinc.json 2> $O/bench_zinc.err; echo "bench zinc rc=$?" >> $O/rc.txt
# same-box A/B of the arithmetic forms (the 6-product bf16 form of rounds 2-3 against the default)
GPS_GEMM_F16=0 GPS_WGRAD_F16=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg > $O/bench_bf16x6.json 2> $O/bench_bf16x6.err; echo "bench bf16x6 rc=$?" >> $O/rc.txt
GPS_MULTIHOT_WGRAD=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg > $O/bench_multihot.json 2> $O/bench_multihot.err; echo "bench multihot rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
    [ $w = pcqm4m ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring_384.txt 2>&1
GEMM_BENCH=256,25600,76800 timeout 300 python tools/gemm_panel_bench.py > $O/gemm_ring_256.txt 2>&1
timeout 300 python tools/wgrad16_probe.py > $O/wgrad16.txt 2>&1
timeout 1200 bash tools/pmc_collect.sh $O/pmc > $O/pmc_collect.log 2>&1
python - <<'PY'
import json
for n in ('bench_default','bench_code2','bench_zinc','bench_bf16x6','bench_multihot','prof_pcqm4m','prof_code2'):
    try:
        d=json.loads(open(f'gpurun_out/r5v/{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('launch_trial_ms'), d.get('pcie_inclusive_ms_per_step'), json.dumps(d.get('roofline'))[:260])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/rc.txt; cat $O/wgrad16.txt; head -12 $O/gemm_ring_384.txt

#!/bin/bash
# Round-4 evidence run, trimmed to ~10 min of box time (27 GPU-minutes were left when the round was re-entered): full GPU
# suite (3 xdist workers, per-file), smoke, the default / code2 / zinc bench lines, kernel traces of the pcqm4m and code2
# bench commands, two PMC passes (FETCH_SIZE, WRITE_SIZE) over tools/kernel_probe.py.  Outputs under gpurun_out/r5g.
set -u
O=gpurun_out/r5g; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider -n 3 --dist loadfile > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
echo "t_pytest=$(( $(date +%s) - T0 ))" >> $O/rc.txt
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
echo "t_bench=$(( $(date +%s) - T0 ))" >> $O/rc.txt
export TMPDIR=/tmp
cd /tmp
for w in pcqm4m code2; do
  rm -rf /tmp/prof_$w
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o bench -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-gemm-tuning > $R/$O/prof_$w.json 2> $R/$O/prof_$w.log
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_stats.py $DB --top 70 > $R/$O/kernel_trace_stats_$w.txt 2>&1
    [ $w = pcqm4m ] && python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$w.txt 2>&1
  fi
  rm -rf /tmp/prof_$w
done
cd $R
echo "t_prof=$(( $(date +%s) - T0 ))" >> $O/rc.txt
timeout 300 python bench.py --workload code2 --no-cpu-baseline > $O/bench_code2.json 2> $O/bench_code2.err; echo "bench code2 rc=$?" >> $O/rc.txt
timeout 200 python bench.py --workload zinc --no-cpu-baseline > $O/bench_zinc.json 2> $O/bench_zinc.err; echo "bench zinc rc=$?" >> $O/rc.txt
echo "t_bench2=$(( $(date +%s) - T0 ))" >> $O/rc.txt
# PMC: HBM bytes of the hand-written kernels (two separate passes, no other trace domains)
mkdir -p $O/pmc
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  GPS_PROBE_GEMM=0 timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o probe -- python $R/tools/kernel_probe.py > $R/$O/pmc/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB --match "k_gatedgcn|k_sattn|k_attn|k_wgrad" > $R/$O/pmc/pmc_$i.txt 2>&1
  rm -rf /tmp/pmc_$i
done
cd $R
python tools/pmc_summarize.py $O/pmc > $O/pmc/summary.json 2> $O/pmc/summary.err
echo "t_all=$(( $(date +%s) - T0 ))" >> $O/rc.txt
python - <<'PY'
import json
for n in ('bench_default','bench_code2','bench_zinc','prof_pcqm4m','prof_code2'):
    try:
        d=json.loads(open(f'gpurun_out/r5g/{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'],3), round(d['value']), d['launch_mode'][:14], d.get('launch_trial_ms'), d.get('pcie_inclusive_ms_per_step'), json.dumps(d.get('roofline'))[:260])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/rc.txt

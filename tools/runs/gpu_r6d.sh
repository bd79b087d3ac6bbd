#!/bin/bash
# Round 5, call 4: gather-sum encoders (csrc/embed.hip) + merged index fills + the core fork on by default: parity, then the
# step A/B, then what sits in the ~90 us gaps at the head of the replayed step (kernel + memory-copy + scratch-memory trace of
# the OLD encoder form) and the timeline of the new one.
set -u
O=gpurun_out/r6d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 300 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -q -p no:cacheprovider -x -k "embed or graph_index or pool or encoders_and_heads or full_model_vs_oracle or code2_model" > $O/pytest_new.log 2>&1; rc=$?; echo "pytest new rc=$rc"
tail -3 $O/pytest_new.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph"
run() { n=$1; shift
  env "$@" timeout 200 $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms  loss {d['final_loss']:.4f}")
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run new A=1
run old_encoders GPS_EMBED_SUM=0
run no_corefork GPS_CORE_FORK=0
run old_both GPS_EMBED_SUM=0 GPS_CORE_FORK=0
run new2 A=1
B_SAVE=$B; B="$B_SAVE --workload code2"
run code2_new A=1
run code2_old GPS_EMBED_SUM=0
B="$B_SAVE --workload zinc"
run zinc_new A=1
run zinc_old GPS_EMBED_SUM=0
B=$B_SAVE
echo "t_ab=$(( $(date +%s) - T0 ))"
export TMPDIR=/tmp; R=$PWD
cd /tmp
for v in old new; do
  rm -rf /tmp/prof_$v
  if [ $v = old ]; then E=0; else E=1; fi
  GPS_EMBED_SUM=$E timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --scratch-memory-trace --stats -d /tmp/prof_$v -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-h2d-leg --no-bucketed-leg --no-secondary --launch graph > $R/$O/prof_$v.json 2> $R/$O/prof_$v.log
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_timeline.py $DB --full > $R/$O/timeline_$v.txt 2>&1
    python $R/tools/rocpd_head.py $DB > $R/$O/head_$v.txt 2>&1
  fi
  rm -rf /tmp/prof_$v
done
cd $R
head -40 $O/head_old.txt
echo "t_trace=$(( $(date +%s) - T0 ))"
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -3 $O/pytest_all.log
echo "t_all=$(( $(date +%s) - T0 ))"

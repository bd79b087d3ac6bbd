#!/bin/bash
set -u
O=gpurun_out/r3o; mkdir -p $O
for m in keep_graph record_only manual_keep loader; do
  timeout 120 python -X faulthandler tools/capture_probe.py $m > $O/$m.log 2>&1
  echo "$m rc=$? ok=$(grep -c 'CAPTURE OK' $O/$m.log)"
done
timeout 600 python -X faulthandler -m pytest tests/test_hip_optim.py -m gpu -q -p no:cacheprovider > $O/pytest_optim.log 2>&1; echo "optim rc=$?"
tail -5 $O/pytest_optim.log
timeout 400 python -X faulthandler bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -E "host-batch|launch-mode|timed region" $O/bench_default.err

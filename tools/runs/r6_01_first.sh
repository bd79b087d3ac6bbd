#!/bin/bash
# round 6, first box: the new parity cases (long graphs, hub, host targets), then the full GPU suite and a bench line
set -u
OUT=gpurun_out/r6_01; mkdir -p $OUT
python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "hub or 4500 or 5000 or malnet or rejects_out_of_range" > $OUT/new_ops.log 2>&1; echo "new ops rc=$?" >> $OUT/summary.txt
python -m pytest tests/test_hip_padding.py -q -x -m gpu -k "host_targets or eval_epoch" > $OUT/new_pad.log 2>&1; echo "new pad rc=$?" >> $OUT/summary.txt
python -m pytest tests/test_hip_layer.py -q -x -m gpu -k "dropout_on" -s > $OUT/layer_dropout.log 2>&1; echo "layer dropout rc=$?" >> $OUT/summary.txt
python bench.py --steps 20 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
python -m pytest tests/ -q -m gpu -x > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?" >> $OUT/summary.txt
tail -3 $OUT/new_ops.log $OUT/new_pad.log $OUT/layer_dropout.log $OUT/gpu_suite.log
cat $OUT/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_01/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','launch_mode','dispatches_per_step')})
print(d['roofline']); print(d['roofline_step'])
PY

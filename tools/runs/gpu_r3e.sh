#!/bin/bash
# round 3: generalised ring GEMM (tests, d = 384 regression check, d = 256 table), embedding gradient, dropout-ON and
# code2 attribution tests, code2 bench line
set -u
O=gpurun_out/r3e; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_norm.py -m gpu -q -p no:cacheprovider -k "gemm or embedding or race" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" > $O/rc.txt
tail -4 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -s -k "dropout_on or code2" > $O/pytest_layer.log 2>&1; echo "layer rc=$?" >> $O/rc.txt
grep -E "passed|failed|dropout on|grad e max|parameter gradients|code2 model|per-graph|ReLU sign|code2 layer" $O/pytest_layer.log | tail -30
timeout 300 python tools/gemm_panel_bench.py > $O/gemm_384.txt 2>&1
GEMM_BENCH=256,25600,76800 timeout 300 python tools/gemm_panel_bench.py > $O/gemm_256.txt 2>&1
grep -v amdgpu $O/gemm_384.txt | head -12; grep -v amdgpu $O/gemm_256.txt | head -12
timeout 600 python bench.py --workload code2 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_code2.json 2> $O/bench_code2.err
GPS_GEMM_PANEL=0 timeout 600 python bench.py --workload code2 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_code2_lib.json 2> $O/bench_code2_lib.err
for t in code2 code2_lib; do echo "== $t: $(python -c "import json; d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('launch_mode'))" 2>&1 | tail -1)"; done
cat $O/rc.txt

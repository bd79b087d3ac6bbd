#!/bin/bash
# round 3: EDGE GEMM with 128-row panels + straight-line stores; XCD-contiguous work items in k_wgrad_stream (A/B)
set -u
O=gpurun_out/r3z; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "gemm or wgrad or race or baseline_sizes" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -5 $O/pytest.log
GPS_GEMM_PANEL=0 timeout 600 python -m pytest tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "baseline_sizes and 304" > $O/pytest_lib304.log 2>&1; echo "lib-GEMM 304 rc=$?"
tail -4 $O/pytest_lib304.log
GEMM_BENCH=304,7569,15348 timeout 300 python tools/gemm_panel_bench.py > $O/gemm_304.txt 2>&1; head -10 $O/gemm_304.txt | tail -9
for i in 1 2; do
for cfg in "GPS_WGRAD_XCD_MAP=1" "GPS_WGRAD_XCD_MAP=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')_$i
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O/bench_$tag.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d.get("in_step_kernel_ms", {})
def mean(sub):
    v = [x["ms"] * 1e3 for k, x in ks.items() if sub in k]
    return round(sum(v) / len(v), 1) if v else None
print(f"== [{sys.argv[2]}] {d['ms_per_step']:.3f} ms roofline {d['roofline']['frac']:.3f} wgrad {mean('k_wgrad_stream')} gg_bwd {mean('k_gatedgcn_bwd')}")
PY
done
done

#!/bin/bash
# round 3: gradients written straight into the optimizer arena; single-stream default schedule
set -u
O=gpurun_out/r3w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "not code2" > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -4 $O/pytest.log
for cfg in "X=1" "GPS_GEMM_STATS=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O/bench_$tag.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d.get("in_step_kernel_ms", {})
def mean(sub):
    v = [x["ms"] * 1e3 for k, x in ks.items() if sub in k]
    return round(sum(v) / len(v), 1) if v else None
print(f"== [{sys.argv[2]}] {d['ms_per_step']:.3f} ms pcie {d.get('pcie_inclusive_ms_per_step')} roofline {d['roofline']['frac']:.3f} gg_fwd {mean('k_gatedgcn_fwd')} gg_bwd {mean('k_gatedgcn_bwd')} mta {mean('multi_tensor')} ")
PY
done

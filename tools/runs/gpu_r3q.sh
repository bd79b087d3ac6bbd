#!/bin/bash
# round 3: schedule A/B -- forked vs serial attention half, GatedGCN launch shape, statistics in / out of the GatedGCN kernel
set -u
O=gpurun_out/r3q; mkdir -p $O
for cfg in "X=1" "GPS_GG_TARGET_WG=256" "GPS_BRANCH_STREAM=0" "GPS_GG_STATS=0" "GPS_GG_STATS=0 GPS_GG_TARGET_WG=256" "GPS_BRANCH_STREAM=0 GPS_GG_STATS=0" "GPS_BRANCH_STREAM=0 GPS_GG_TARGET_WG=256"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - "$O/bench_$tag.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d.get("in_step_kernel_ms", {})
def mean(sub):
    v = [x["ms"] * 1e3 for k, x in ks.items() if sub in k]
    return round(sum(v) / len(v), 1) if v else None
print(f"== [{sys.argv[2]}] {d['ms_per_step']:.3f} ms  gg_fwd {mean('k_gatedgcn_fwd')} gg_bwd {mean('k_gatedgcn_bwd')} sattn_fwd {mean('k_sattn_fwd')} sattn_bwd {mean('k_sattn_bwd')} rows {mean('k_rows_fwd')} ring13 {mean('k_gemm_ring<1, 3, 3')}")
PY
done

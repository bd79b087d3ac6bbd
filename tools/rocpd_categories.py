#!/usr/bin/env python
"""Per-step GPU time by category from a rocprofv3 rocpd database of `bench.py --steps K --warmup W`.

    python tools/rocpd_categories.py gpurun_out/prof/bench_results.db --steps 13
"""
import argparse
import re
import sqlite3

CATS = [
    ("dense GEMM (rocBLAS/hipBLASLt)", r"^Cijk_|gemm|Gemm"),
    ("weight-gradient split-K MFMA (HIP)", r"k_wgrad"),
    ("segment attention (HIP)", r"k_attn_"),
    ("GatedGCN / GINE sparse (HIP)", r"k_gatedgcn|k_gine"),
    ("FAVOR+ (HIP)", r"k_favor"),
    ("BN / residual / dropout task lists (HIP)", r"k_bn_|k_act_drop|k_colsum|k_rows_fwd|k_stats_finalize|k_bwd_partial|k_bwd_finalize|k_bwd_apply"),
    ("graph index + pooling (HIP)", r"k_histogram|k_scan|k_fill|k_sort_and_resolve|k_tile_map|k_ptr_from|k_pool|k_node_graph|k_segment_max"),
    ("clip + AdamW (HIP) + gradient pack", r"k_adamw|k_sqnorm|multi_tensor|FusedOptimizer|lpnorm|LpNorm"),
    ("ATen elementwise / reduce / copy", r"elementwise|reduce_kernel|CatArray|copyBuffer|fillBuffer|index|scatter|gather|embedding|rocprim|sort"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, required=True, help="steps + warm-up steps executed")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
    tot = sum(r[2] for r in rows)
    acc = {k: [0, 0.0] for k, _ in CATS}
    acc["other"] = [0, 0.0]
    for name, n, t in rows:
        for k, pat in CATS:
            if re.search(pat, name):
                acc[k][0] += n; acc[k][1] += t
                break
        else:
            acc["other"][0] += n; acc["other"][1] += t
    print(f"# {a.db}: {tot/1e6/a.steps:.2f} ms GPU kernel time and {sum(r[1] for r in rows)/a.steps:.0f} launches per step")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{t/tot*100:6.1f}%  {t/1e6/a.steps:7.3f} ms/step  {n/a.steps:7.1f} launches/step  {k}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Launch shapes of every kernel in a rocprofv3 rocpd database: workgroups, threads, registers, LDS and -- from those -- the
workgroups a CU can hold and the DISPATCH ROUNDS the launch makes on the chip (a round that is mostly empty is time lost:
the GatedGCN launches and the FAVOR+ context slices of round 6 were found this way).

    python tools/rocpd_grids.py /tmp/prof/bench_results.db [--cus 256] [--top 40] [--objs graphgps_amd/csrc/*.o]

Registers: the tracer's `vgpr_count` under-reports kernels without accumulation registers (k_sattn_bwd<24>: 124 there, 251 in
the code object -- an LDS diet that would have made room for a fourth workgroup at 124 changed nothing: the kernel sits at two
per CU); with --objs the allocation comes from the code objects' metadata (tools/kernel_regs.py) instead.
"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--objs", nargs="*", default=[])
    a = ap.parse_args()
    meta = {}
    if a.objs:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kernel_regs
        for o in a.objs:
            try:
                for r in kernel_regs.kernels(o):
                    meta[r["pretty"]] = (r["vgpr"], r["agpr"])
            except Exception as exc:          # noqa: BLE001
                print(f"# {o}: {exc}")
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    g = lambda *names: next((n for n in names if n in cols), None)
    gx, wx = g("grid_size_x", "grid_x", "grid_size"), g("workgroup_size_x", "workgroup_x", "workgroup_size")
    gy, gz, wy, wz = g("grid_size_y", "grid_y"), g("grid_size_z", "grid_z"), g("workgroup_size_y", "workgroup_y"), g("workgroup_size_z", "workgroup_z")
    ag = g("accum_vgpr_count", "agpr_count")
    if not gx or not wx:
        print("columns:", cols)
        return
    expr_g = f"{gx}" + (f"*{gy}*{gz}" if gy and gz else "")
    expr_w = f"{wx}" + (f"*{wy}*{wz}" if wy and wz else "")
    q = (f"select name, {expr_g}, {expr_w}, max(vgpr_count), {('max(' + ag + ')') if ag else '0'}, max(lds_size), count(*), sum(end-start), "
         f"avg(end-start) from kernels group by name, {expr_g}, {expr_w} order by 8 desc")
    rows = c.execute(q).fetchall()
    print(f"{'total_us':>10} {'avg_us':>8} {'calls':>6} {'wgs':>6} {'thr':>5} {'vgpr':>5} {'agpr':>5} {'lds':>7} {'wg/cu':>6} {'rounds':>7}  name")
    for name, grid, wg, vg, agc, lds, n, tot, avg in rows[:a.top]:
        from_meta = name in meta
        if from_meta:
            vg, agc = meta[name]          # (.vgpr_count of the metadata is the unified total, accumulation registers included)
        wgs = grid // wg if wg else 0
        waves = (wg + 63) // 64
        regs = ((vg or 0) + (0 if from_meta else (agc or 0)) + 7) // 8 * 8
        per_simd = 512 // regs if regs else 8
        per_simd = min(per_simd, 8)
        by_regs = (per_simd * 4) // waves if waves else 0
        by_lds = (160 * 1024) // lds if lds else 99
        by_thr = 2048 // wg if wg else 0
        fit = max(1, min(by_regs, by_lds, by_thr))
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        m = re.search(r"(k_\w+(?:<[^>]*>)?|Cijk_\w{0,20}|[\w:]+)", short.replace("void ", ""))
        print(f"{tot/1e3:10.1f} {avg/1e3:8.2f} {n:6d} {wgs:6d} {wg:5d} {vg or 0:5d} {agc or 0:5d} {lds or 0:7d} {fit:6d} {wgs / (fit * a.cus):7.2f}  {(m.group(1) if m else short)[:70]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into the per-kernel table
`rocprofv3 --stats` would print: calls, total/avg/min/max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db [--top 60] [--match REGEX]
"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--match", default=None)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                     "max(end-start), max(vgpr_count), max(lds_size) from kernels group by name "
                     "order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {a.db}: {sum(r[1] for r in rows)} kernel dispatches, {tot/1e6:.3f} ms total GPU kernel time")
    print(f"{'%':>6} {'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>8} {'max_us':>9} {'vgpr':>5}  name")
    shown = 0
    for name, n, s, avg, mn, mx, vg, lds in rows:
        if a.match and not re.search(a.match, name):
            continue
        print(f"{s/tot*100:6.2f} {n:7d} {s/1e3:11.1f} {avg/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:9.2f} "
              f"{vg or 0:5d}  {name[:150]}")
        shown += 1
        if shown >= a.top:
            break


if __name__ == "__main__":
    main()

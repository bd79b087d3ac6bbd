#!/usr/bin/env python
"""Per-kernel mean of each PMC counter in a rocprofv3 rocpd database.

    python tools/rocpd_pmc.py out/probe_results.db [--match REGEX]
"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="k_")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {namecol}, counter_name, count(*), avg(value), min(value), max(value) "
                     f"from counters_collection group by {namecol}, counter_name").fetchall()
    print(f"# {a.db}  (columns: {cols})")
    for name, ctr, n, avg, mn, mx in rows:
        if a.match and not re.search(a.match, name):
            continue
        print(f"{ctr:28s} n={n:5d} avg={avg:16.1f} min={mn:16.1f} max={mx:16.1f}  {name[:110]}")


if __name__ == "__main__":
    main()

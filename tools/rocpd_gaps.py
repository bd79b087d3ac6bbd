#!/usr/bin/env python
"""Idle time between consecutive kernels per HIP stream/queue in a rocprofv3 rocpd database: how much of a
step is launch gap rather than kernel time.

    python tools/rocpd_gaps.py out/bench_results.db --steps 16
"""
import argparse
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--steps", type=int, default=1)
a = ap.parse_args()
c = sqlite3.connect(a.db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
print("# columns:", cols)
rows = c.execute(f"select {key}, start, end, name from kernels order by start").fetchall()
by = defaultdict(list)
for k, s, e, n in rows:
    by[k].append((s, e, n))
for k, ks in by.items():
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 < g < 100_000]
    hist = defaultdict(int)
    for g in small:
        hist[min(int(g // 1000), 20)] += 1
    print(f"{key}={k}: {len(ks)/a.steps:.0f} kernels/step, busy {busy/1e6/a.steps:.2f} ms/step, "
          f"gaps<100us: {sum(small)/1e6/a.steps:.2f} ms/step over {len(small)/a.steps:.0f} gaps "
          f"(mean {sum(small)/max(len(small),1)/1e3:.2f} us); overlap(neg gaps) {sum(1 for g in gaps if g<=0)/a.steps:.0f}/step")
    print("   gap histogram (us: count/step):", {f"{b}": round(n / a.steps, 1) for b, n in sorted(hist.items())})

# ---- GPU-wide idle time: union of all kernel intervals vs the span they cover -------------------
iv = sorted((s, e) for s, e, _ in ((r[1], r[2], 0) for r in rows))
# keep only the steady-state tail: last 60 % of dispatches (skips tuning / capture / warm-up)
iv = iv[int(len(iv) * 0.4):]
span = iv[-1][1] - iv[0][0]
covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
idle_hist = defaultdict(int)
idle_ns = defaultdict(int)
for s, e in iv[1:]:
    if s > cur_e:
        covered += cur_e - cur_s
        g = s - cur_e
        b = min(int(g // 2000) * 2, 40)
        idle_hist[b] += 1
        idle_ns[b] += g
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
covered += cur_e - cur_s
print(f"GPU-wide (steady-state tail): span {span/1e6:.2f} ms, some kernel running {covered/1e6:.2f} ms "
      f"({100*covered/span:.1f} %), idle {100*(span-covered)/span:.1f} %")
print("   idle-gap histogram (us bucket: count, total ms):",
      {f"{b}-{b+2}": (n, round(idle_ns[b] / 1e6, 2)) for b, n in sorted(idle_hist.items())})

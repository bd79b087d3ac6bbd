#!/usr/bin/env python
"""What a rocprofv3 rocpd database knows about the dispatches of one step that follow an idle gap: every column of the
kernels view for them (scratch / private-segment sizes, grid, LDS), plus the tables / views the database holds and every
memory-copy and scratch-memory record inside the step's window.  Written to find out what sits in the ~90 us gaps in front of
four ATen index kernels at the head of the replayed pcqm4m step."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables / views:", ", ".join(n for n in names if not n.startswith("rocpd_info")))
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("# kernels columns:", ", ".join(cols))
rows = c.execute("select * from kernels order by start").fetchall()
ix = {k: i for i, k in enumerate(cols)}
marks = [i for i, r in enumerate(rows) if "k_adamw" in str(r[ix["name"]])]
if len(marks) < 3:
    raise SystemExit("fewer than 3 optimizer launches")
lo, hi = marks[-3] + 1, marks[-2] + 1
step = rows[lo:hi]
t0 = step[0][ix["start"]]
show = [k for k in cols if any(s in k.lower() for s in ("scratch", "private", "lds", "grid", "workgroup", "vgpr", "sgpr"))]
print("# dispatches of one step behind an idle gap > 30 us (and their predecessor):")
prev_end = None
for j, r in enumerate(step):
    st, en = r[ix["start"]], r[ix["end"]]
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    if gap > 30.0 or j < 3:
        for q in (step[j - 1], r) if j else (r,):
            print(f"{(q[ix['start']] - t0) / 1e3:9.1f} us  +{(q[ix['end']] - q[ix['start']]) / 1e3:6.1f}  gap {gap:6.1f}  "
                  + str(q[ix['name']])[:70] + "  " + " ".join(f"{k}={q[ix[k]]}" for k in show))
        print()
    prev_end = en
w0, w1 = step[0][ix["start"]], step[-1][ix["end"]]
for view in ("memory_copies", "scratch_memory", "memory_allocations"):
    if view in names:
        vc = [r[1] for r in c.execute(f"pragma table_info({view})")]
        try:
            rs = c.execute(f"select * from {view} where start >= {w0} and start <= {w1} order by start").fetchall()
        except sqlite3.Error as exc:
            print(f"# {view}: {exc}")
            continue
        print(f"# {view}: {len(rs)} records inside the step; columns {vc}")
        for r in rs[:40]:
            d = dict(zip(vc, r))
            print("   ", {k: (round((v - t0) / 1e3, 1) if k in ("start", "end") else v) for k, v in d.items()
                          if k in ("start", "end", "name", "size", "src_agent_abs_index", "dst_agent_abs_index", "operation",
                                   "alloc_flags", "kind")})

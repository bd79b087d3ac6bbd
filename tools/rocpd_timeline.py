#!/usr/bin/env python
"""One steady-state training step out of a rocprofv3 rocpd database as a timeline: every kernel dispatch of the step with
its stream, start offset, duration and the idle gap in front of it; then the per-layer view (median duration of the
k-th launch of a layer over the layers of the step) and where the GPU was idle.

    python tools/rocpd_timeline.py out/bench_results.db [--step -2] [--full]

A step is delimited by the optimizer kernel (k_adamw).  Names are shortened to the kernel's own name + template args.
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]*>)?)", n)
    s = m.group(1) if m else n
    if s.startswith("Cijk"):
        s = "Cijk(" + (re.search(r"MT\d+x\d+x\d+", n) or [""])[0] + ")"
    return s[:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step", type=int, default=-2, help="which step (index into the list of steps; -2 = last full one)")
    ap.add_argument("--full", action="store_true", help="print every dispatch of the step")
    ap.add_argument("--layers", type=int, default=10)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    rows = c.execute(f"select {key}, start, end, name from kernels").fetchall()
    try:        # copies / fills that are not kernels (rocprofv3 --memory-copy-trace): they explain idle gaps
        rows += [(r[0], r[1], r[2], f"memcpy:{r[3]}:{r[4]}B") for r in
                 c.execute(f"select {key}, start, end, name, size from memory_copies")]
    except sqlite3.Error:
        pass
    rows.sort(key=lambda r: r[1])
    marks = [i for i, r in enumerate(rows) if "k_adamw" in r[3]]
    if len(marks) < 3:
        raise SystemExit("fewer than 3 optimizer launches in the trace")
    lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    print(f"# step of {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us from first start to last end")
    streams = sorted({r[0] for r in step})
    sid = {s: i for i, s in enumerate(streams)}
    if a.full:
        last_end = {}
        for s, st, en, n in step:
            gap = (st - last_end[s]) / 1e3 if s in last_end else 0.0
            last_end[s] = en
            print(f"{(st - t0) / 1e3:9.1f} +{(en - st) / 1e3:7.1f} us  s{sid[s]}  gap {gap:6.1f}  {short(n)}")
    # GPU-wide busy / idle inside the step
    iv = sorted((st, en) for _, st, en, _ in step)
    cur_s, cur_e = iv[0]
    busy, idles = 0, []
    for st, en in iv[1:]:
        if st > cur_e:
            busy += cur_e - cur_s
            idles.append(st - cur_e)
            cur_s, cur_e = st, en
        else:
            cur_e = max(cur_e, en)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0] if iv else 1
    print(f"# some kernel running {busy / 1e3:.1f} us of {span / 1e3:.1f} ({100 * busy / span:.1f} %); "
          f"{len(idles)} idle gaps, total {sum(idles) / 1e3:.1f} us, mean {sum(idles) / max(len(idles), 1) / 1e3:.2f} us")
    # per-kernel totals in this step
    tot = defaultdict(lambda: [0, 0])
    for _, st, en, n in step:
        k = short(n)
        tot[k][0] += 1
        tot[k][1] += en - st
    print("# per kernel in this step: calls, total us, mean us")
    for k, (cnt, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {cnt:4d} {t / 1e3:9.1f} {t / cnt / 1e3:8.2f}  {k}")
    # exclusive time: for each instant only the kernel that started first counts (critical-path flavour)
    ev = sorted(step, key=lambda r: r[1])
    excl = defaultdict(int)
    t = ev[0][1]
    for _, st, en, n in ev:
        if en <= t:
            continue
        excl[short(n)] += en - max(st, t)
        t = max(t, en)
    print("# time not covered by an earlier-started kernel (sums to the busy time):")
    for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:30]:
        print(f"  {v / 1e3:9.1f}  {k}")


if __name__ == "__main__":
    main()

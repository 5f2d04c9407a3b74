#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as text:
per-kernel totals and per launch-shape averages.  usage: prof_summary.py results.db [steps]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    print(f"# kernel totals (all dispatches in the run; {steps:g} steps incl. warm-up); total GPU kernel time "
          f"{total / 1e6:.3f} ms = {total / 1e6 / steps:.3f} ms/step")
    print(f"{'calls':>7} {'total_ms':>10} {'ms/step':>9} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}  name")
    for n, c, s, a, mn, mx in rows:
        print(f"{c:7d} {s / 1e6:10.3f} {s / 1e6 / steps:9.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} "
              f"{100 * s / total:6.2f}  {short(n)}")
    print("\n# per launch shape (grid in threads) for the GEMM / conv / attention kernels")
    rows = list(cur.execute(
        "select name, grid_x, grid_y, count(*), avg(duration), sum(duration) from kernels "
        "where name like '%gemm_%' or name like '%attn_%' or name like '%conv_ll%' or name like '%stem5%' or name like '%_rr_kernel%' or name like '%fused_kernel%' "
        "group by name, grid_x, grid_y order by sum(duration) desc"))
    for n, gx, gy, c, a, s in rows:
        print(f"{c:7d} {s / 1e6 / steps:9.3f} ms/step {a / 1e3:9.2f} us  grid=({gx},{gy})  {short(n)}")
    concurrency(db, cur)


def concurrency(db, cur):
    """GPU occupancy of the stream timeline: union of kernel intervals vs the sum of their durations."""
    rows = list(cur.execute("select start, end from kernels order by start"))
    if not rows:
        return
    # skip the start-up part (weight upload / first-step allocation): keep the last 60 % of the dispatches
    rows = rows[int(len(rows) * 0.4):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cs, ce = 0, rows[0][0], rows[0][1]
    for a, b in rows[1:]:
        if a > ce:
            busy += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    total = sum(b - a for a, b in rows)
    gaps(cur, rows[0][0])
    print(f"\n# timeline over the last 60% of dispatches: wall {1e-6 * (t1 - t0):.2f} ms, some kernel running "
          f"{100.0 * busy / (t1 - t0):.1f}% of it, sum of kernel durations / wall = {total / (t1 - t0):.2f} "
          f"(average kernels in flight)")


def gaps(cur, t_from):
    """Idle gaps of the device (no kernel of any stream running): histogram, and the (kernel before -> kernel after) pairs
    that own most of the idle time."""
    rows = list(cur.execute("select start, end, name from kernels where start >= ? order by start", (t_from,)))
    if len(rows) < 2:
        return
    edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 1 << 30]
    hist = [[0, 0.0] for _ in edges]
    pairs = {}
    ce, cname = rows[0][1], rows[0][2]
    for a, b, n in rows[1:]:
        if a > ce:
            g = (a - ce) / 1e3
            for i in range(len(edges) - 1):
                if g < edges[i + 1]:
                    hist[i][0] += 1
                    hist[i][1] += g
                    break
            k = (short(cname)[:60], short(n)[:60])
            c = pairs.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += g
        if b > ce:
            ce, cname = b, n
    tot = sum(h[1] for h in hist)
    print(f"\n# idle gaps (no kernel running) over the same window: {sum(h[0] for h in hist)} gaps, {tot / 1e3:.3f} ms")
    for i in range(len(edges) - 1):
        hi = "inf" if edges[i + 1] >= 1 << 30 else str(edges[i + 1])
        print(f"  {edges[i]:>4} .. {hi:>4} us: {hist[i][0]:6d} gaps {hist[i][1] / 1e3:9.3f} ms")
    print("# idle time by (kernel that ended last -> kernel that started next), top 25")
    for (a, b), (c, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {c:6d} x {g / c:8.2f} us = {g / 1e3:8.3f} ms   {a}  ->  {b}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the time between kernels goes: reads a rocprofv3 kernel trace (…_kernel_trace.csv), keeps the library's
kernels, and prints busy time, idle time and the idle time by (previous kernel -> next kernel) transition.
usage: trace_gaps.py [--by-grid] <kernel_trace.csv> [name-filter]   (--by-grid: one line per kernel and grid)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    rows = []
    by_grid = "--by-grid" in sys.argv
    if by_grid:
        sys.argv.remove("--by-grid")
    flt = sys.argv[2] if len(sys.argv) > 2 else "srx::"
    for r in csv.DictReader(open(sys.argv[1])):
        if flt in r["Kernel_Name"]:
            name = short(r["Kernel_Name"])
            if by_grid:
                name += f' [{int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])}x{int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])}]'
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} launches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms (sum of durations)")
    per = defaultdict(lambda: [0, 0, 0, 0])
    for s, e, n in rows:
        d = e - s
        per[n][0] += 1
        per[n][1] += d
        if d < 2500:
            per[n][2] += 1
            per[n][3] += d
    print("kernel: launches, total ms | of them shorter than 2.5 us: launches, total ms")
    for n, (c, t, c2, t2) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:46s} {c:6d} {t / 1e6:8.2f} | {c2:6d} {t2 / 1e6:7.3f}")
    gaps = defaultdict(lambda: [0, 0])
    hist = defaultdict(int)
    covered_until = rows[0][1]
    idle = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - covered_until
        if g > 0:
            idle += g
            key = (n0, n1) if g < 20000 else ("(long gap)", n1)
            gaps[key][0] += g
            gaps[key][1] += 1
            b = 1
            while b < g / 1000:
                b *= 2
            hist[b] += 1
        covered_until = max(covered_until, e1)
    print(f"idle between kernels {idle / 1e6:.2f} ms = {100 * idle / span:.1f} % of the span")
    print("gap histogram (upper bound us: count):", dict(sorted(hist.items())))
    for (a, b), (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {t / 1e3:9.1f} us in {n:5d} gaps ({t / n / 1e3:6.2f} us each)  {a} -> {b}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The solver's two kernels over one step of the benchmark, iteration by iteration, and the flow kernels that ran
beside them: reads a rocprofv3 kernel trace of `bench.py`, takes the last complete solve (180 forward launches) and
prints per group of iterations the mean forward and update durations, the gap between the two, and how many flow
kernels overlapped (with their summed duration).
usage: trace_timeline.py <kernel_trace.csv> [iterations-per-group]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "srx::" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fwd = [i for i, r in enumerate(rows) if "k_forward_sign" in r[2]]
upd = [i for i, r in enumerate(rows) if "k_backproject_update" in r[2]]
iters = 180
f_idx = fwd[-iters:]
t_begin, t_end = rows[f_idx[0]][0], rows[upd[-1]][1]
flow = [r for r in rows if ("tvl1" in r[2] or "k_warp" in r[2] or "k_settle" in r[2] or "k_phase_gate" in r[2] or
                            "k_centered" in r[2] or "k_upsample" in r[2] or "k_pyr" in r[2] or "guarded" in r[2])
        and r[1] > t_begin and r[0] < t_end]
print(f"step span {(t_end - t_begin) / 1e6:.2f} ms, {len(flow)} flow launches inside, flow busy "
      f"{sum(e - s for s, e, _ in flow) / 1e6:.2f} ms, first {(flow[0][0] - t_begin) / 1e6 if flow else 0:.2f} ms, "
      f"last ends {(flow[-1][1] - t_begin) / 1e6 if flow else 0:.2f} ms after the step's start")
u_idx = [i for i in upd if rows[i][0] >= t_begin]
print("iterations  fwd_us  upd_us  gap_fu_us gap_uf_us  flow_launches flow_busy_us")
for g in range(0, iters, grp):
    fs = [rows[i] for i in f_idx[g:g + grp]]
    us = [rows[i] for i in u_idx[g:g + grp]]
    fd = sum(e - s for s, e, _ in fs) / len(fs) / 1e3
    ud = sum(e - s for s, e, _ in us) / len(us) / 1e3
    gfu = sum(u[0] - f[1] for f, u in zip(fs, us)) / len(fs) / 1e3
    nxt = [rows[i] for i in f_idx[g + 1:g + grp + 1]]
    guf = sum(f[0] - u[1] for u, f in zip(us, nxt)) / max(1, len(nxt)) / 1e3
    a, b = fs[0][0], us[-1][1]
    fl = [r for r in flow if r[1] > a and r[0] < b]
    print(f"{g:4d}-{g + grp - 1:<4d}  {fd:7.1f} {ud:7.1f} {gfu:8.1f} {guf:8.1f}   {len(fl):6d} {sum(e - s for s, e, _ in fl) / 1e3:10.1f}")

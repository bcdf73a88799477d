"""Keep the library's own kernels (names in srx::) of a rocprofv3 *kernel_stats.csv and recompute the percentages
over them: the traced process also synthesises its clip with torch, whose kernels are not part of the workload.

usage: filter_kernel_stats.py <in.csv> <out.csv>
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
own = [r for r in rows if "srx::" in r["Name"]]
total = sum(int(r["TotalDurationNs"]) for r in own) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    for r in own:
        r = dict(r)
        r["Percentage"] = round(100.0 * int(r["TotalDurationNs"]) / total, 2)
        for k in ("Calls", "TotalDurationNs", "MinNs", "MaxNs"):
            r[k] = int(r[k])
        for k in ("AverageNs", "StdDev"):
            r[k] = float(r[k])
        w.writerow(r)
print(f"{len(own)} of {len(rows)} kernels kept; {total / 1e6:.1f} ms of library kernels")

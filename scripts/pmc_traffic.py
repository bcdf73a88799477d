"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> JSON.

usage: pmc_traffic.py <fetch pass dir> <write pass dir> <out.json> [workload description]
FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte read requests of
wide coalesced loads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1]
            acc[n].append(float(r['Counter_Value']))
    return {n: sum(v) / len(v) for n, v in acc.items()}


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for n in sorted(set(fetch) & set(write)):
    if 'srx' not in n and not n.startswith('k_'):
        continue
    out[n] = {
        "FETCH_SIZE_KiB": fetch[n],
        "WRITE_SIZE_KiB": write[n],
        "traffic_bytes_per_launch": 2 * fetch[n] * 1024 + write[n] * 1024,
    }
json.dump({"note": "mean per launch; traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction)",
           "workload": sys.argv[4] if len(sys.argv) > 4 else
           "960x540 RGB -> 3840x2160, window 9 (the solver launches of the command the passes were run on)", "kernels": out},
          open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))

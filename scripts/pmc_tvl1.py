"""HBM-side traffic of the TV-L1 stage per frame pair from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> JSON.

usage: pmc_tvl1.py <fetch pass dir> <write pass dir> <pairs in the profiled run> <out.json>
Sums every launch of the library's flow kernels (k_tvl1_*, k_warp, k_pyr_down, k_centered_gradient, k_to_gray,
k_upsample_flow, k_copy16) over the run and divides by the number of pairs; FETCH_SIZE is doubled (gfx950
correction), both are in KiB.
"""
import collections
import csv
import glob
import json
import sys

KERNELS = ("k_tvl1", "k_warp", "k_pyr_down", "k_centered_gradient", "k_to_gray", "k_upsample_flow", "k_copy16", "k_settle",
           "k_phase_gate", "k_zero_guarded", "k_merge2", "k_boundary", "k_open_level")


def totals(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1]
            if n.startswith(KERNELS):
                acc[n][0] += float(r['Counter_Value'])
                acc[n][1] += 1
    return acc


fetch, write = totals(sys.argv[1], 'FETCH_SIZE'), totals(sys.argv[2], 'WRITE_SIZE')
pairs = int(sys.argv[3])
kernels = {}
total = 0.0
for n in sorted(set(fetch) | set(write)):
    b = (2 * fetch[n][0] + write[n][0]) * 1024 / pairs
    kernels[n] = {"launches_per_pair": max(fetch[n][1], write[n][1]) / pairs, "traffic_bytes_per_pair": b}
    total += b
json.dump({"note": "per frame pair; traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction), all flow kernels",
           "workload": "TV-L1 flow on 1920x1080 grey pairs (the command the passes were run on)",
           "pairs": pairs, "traffic_bytes_per_pair": total, "kernels": kernels}, open(sys.argv[4], 'w'), indent=1)
print(json.dumps({"traffic_bytes_per_pair": total}, indent=1))

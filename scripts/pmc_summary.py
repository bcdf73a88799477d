"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name.
usage: pmc_summary.py <output dir> [kernel name substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
pats = sys.argv[2:]
for f in glob.glob(d + '/*/*counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'srx' not in n: continue
        n = n.replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1]
        if pats and not any(p in n for p in pats): continue
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for n, cs in acc.items():
        print(n, {c: round(sum(v)/len(v)) for c, v in cs.items()}, 'calls', len(next(iter(cs.values()))))

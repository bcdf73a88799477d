"""Print per-kernel averages from a rocprofv3 --kernel-trace --stats csv directory."""
import csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if 'srx' in r['Name']:
                n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1]
                print(f"{n:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")

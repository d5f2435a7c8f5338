"""Print per-kernel call counts and average durations from a rocprofv3 kernel_stats.csv under a directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'at::' in n or 'rocclr' in n:
        continue
    print(f"{n.split('(')[0][:48]:48s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} us")

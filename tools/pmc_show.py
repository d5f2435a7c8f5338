"""Average per-dispatch counter values of kernels matching a substring: python tools/pmc_show.py dir substring"""
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if sys.argv[2] not in k:
            continue
        agg[(k[:40], r['Counter_Name'])] += float(r['Counter_Value'])
        cnt[(k[:40], r['Counter_Name'])] += 1
    for k, v in sorted(agg.items()):
        print(k[0], k[1], f"{v / cnt[k]:.4g}", f"(x{cnt[k]})")

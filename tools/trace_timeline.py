"""Timeline of a rocprofv3 --kernel-trace CSV: per kernel start (us since the first kernel of the window), duration and
queue, for the last `count` kernels before the end; and the fraction of the window in which accumulate_kernel runs.
python tools/trace_timeline.py <dir> [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-count:]
t0 = int(rows[0]['Start_Timestamp'])
acc = 0
for r in rows:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkhip::', '')[:28]
    if 'accumulate' in name:
        acc += e - s
    print(f"{s/1e3:10.1f} {(e-s)/1e3:8.1f} q{r['Queue_Id']:>3s} {name}")
span = int(rows[-1]['End_Timestamp']) - t0
print(f"window {span/1e3:.1f} us, accumulate busy {acc/1e3:.1f} us = {acc/span:.2f}")

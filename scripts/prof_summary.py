"""Summarises a rocprofv3 --kernel-trace --stats CSV directory: python scripts/prof_summary.py DIR [iters]"""
import csv, glob, sys
d = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6/iters:.3f} ms per iteration ({iters} iterations)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print(f"{r['Name'][:84]:84s} calls/it {int(r['Calls'])/iters:6.1f} ms/it {float(r['TotalDurationNs'])/1e6/iters:7.3f} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")

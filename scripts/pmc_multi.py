"""Per-kernel sums of several rocprofv3 --pmc counters: python scripts/pmc_multi.py DIR OUT.csv COUNTER [COUNTER ...]
(DIR holds *_counter_collection.csv).  Kernel names are cut at their argument list."""
import csv, glob, sys
from collections import defaultdict

d, out, counters = sys.argv[1], sys.argv[2], sys.argv[3:]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] not in counters:
        continue
    name = r["Kernel_Name"].split("(")[0]
    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
with open(out, "w") as o:
    o.write("kernel,launches," + ",".join(counters) + "\n")
    for name in sorted(tot, key=lambda n: -tot[n][counters[0]]):
        o.write(f"\"{name}\",{len(disp[name])}," + ",".join(f"{tot[name][c]:.0f}" for c in counters) + "\n")
print("wrote", out, "from", f)

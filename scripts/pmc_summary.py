"""Sums a rocprofv3 --pmc counter per kernel name: python scripts/pmc_summary.py DIR COUNTER OUT.csv
(DIR holds *_counter_collection.csv; OUT rows: kernel, launches, <COUNTER>_KB_total, <COUNTER>_KB_per_launch --
FETCH_SIZE / WRITE_SIZE are reported in KB).  The kernel name is cut at its argument list."""
import csv, glob, sys
from collections import defaultdict

d, counter, out = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot, disp = defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"].split("(")[0]
    tot[name] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
with open(out, "w") as o:
    o.write(f"kernel,launches,{counter}_KB_total,{counter}_KB_per_launch\n")
    for name in sorted(tot, key=lambda n: -tot[n]):
        n = len(disp[name])
        o.write(f"\"{name}\",{n},{tot[name]:.0f},{tot[name] / n:.1f}\n")
print("wrote", out, "from", f)

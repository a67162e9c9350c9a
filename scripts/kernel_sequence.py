"""Kernel sequence of the last building of a rocprofv3 --kernel-trace run, in start order:
  python scripts/kernel_sequence.py TRACE_DIR [n_steps_in_run] > sequence.txt
Columns: start (us since the step's first kernel), duration (us), gap to the previous kernel of the same queue (us),
queue, name."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the step = from the last k_vox / voxelize kernel group to the end
starts = [i for i, r in enumerate(rows) if "k_insert_points" in r["Kernel_Name"]]
i0 = starts[-1]
# walk back to the voxelize kernels just before
while i0 > 0 and "k_vox" in rows[i0 - 1]["Kernel_Name"]:
    i0 -= 1
seq = rows[i0:]
t0 = int(seq[0]["Start_Timestamp"])
last_end = {}
for r in seq:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  q{q}  {r['Kernel_Name'][:110]}")

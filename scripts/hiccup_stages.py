"""Development probe: 300 passes with the timeline marks on; for the passes that take over 1.25 x the median, where the
time went (host clock and stream time of every stage against the median pass)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd import timeline
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
clouds = [torch.from_numpy(make_scene(i, 500000)).to(dev) for i in range(4)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
passes = []
with torch.no_grad():
    for i in range(8):
        model(list(voxelize(clouds[i % 4], 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)))
    torch.cuda.synchronize()
    for rep in range(N):
        timeline.MARKS = []
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        h0 = time.perf_counter()
        inp = list(voxelize(clouds[rep % 4], 50, cfg.SPARSE3D.VOXEL_FULL_SCALE))
        hv = time.perf_counter()
        model(inp)
        h1 = time.perf_counter()
        marks, timeline.MARKS = timeline.MARKS, None
        torch.cuda.synchronize()
        row = {"voxelize (host)": (hv - h0) * 1e3, "pass (host)": (h1 - h0) * 1e3}
        for label, k, ev, host in marks:
            key = f"{label} L{k}" if k >= 0 else label
            if ev is None:
                row[key + " (host)"] = (host - h0) * 1e3
            else:
                row[key] = start.elapsed_time(ev)
                row[key + " [enq]"] = (host - h0) * 1e3
        passes.append(row)
tot = sorted(p["pass (host)"] for p in passes)
med = tot[len(tot) // 2]
keys = list(passes[0].keys())
medrow = {k: sorted(p.get(k, 0.0) for p in passes)[len(passes) // 2] for k in keys}
print(f"median pass {med:.3f} ms; slow passes (> 1.25 x): {[round(p['pass (host)'], 2) for p in passes if p['pass (host)'] > 1.25 * med]}")
for p in passes:
    if p["pass (host)"] > 1.25 * med:
        print(f"--- pass of {p['pass (host)']:.2f} ms: first stage that is late by > 0.5 ms, then all lateness")
        late = [(k, p.get(k, 0.0) - medrow[k]) for k in keys]
        late.sort(key=lambda kv: medrow[kv[0]])
        for k, d in late:
            if abs(d) > 0.3:
                print(f"      {k:45s} median {medrow[k]:6.2f}  this pass {p.get(k, 0.0):6.2f}  (+{d:.2f})")

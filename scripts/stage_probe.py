"""Development probe: host wall time of the three pipeline stages, serial and inside serving.BuildingPipeline."""
import os, sys, time, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.serving import BuildingPipeline
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd import voxelize as vz

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
scenes = [torch.from_numpy(make_scene(i, 500000)).to(dev) for i in range(4)]
T = collections.defaultdict(list)


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T[name].append((time.perf_counter() - t0) * 1e3)
        return r
    return w


import detection_3d_amd.serving as serving
serving.voxelize = timed("voxelize", vz.voxelize)
model.stage_geometry = timed("geometry", model.stage_geometry)
model.stage_features = timed("features", model.stage_features)
model.stage_tail = timed("tail", model.stage_tail)


def report(tag, n, dt):
    print(f"{tag}: {n / dt:.1f} buildings/s; host ms per building: " +
          ", ".join(f"{k} {sum(v) / len(v):.2f} (max {max(v):.2f})" for k, v in T.items()), flush=True)
    T.clear()


with torch.no_grad():
    s = cfg.SPARSE3D
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            c, f = serving.voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
            model.stage_tail(model.stage_features(model.stage_geometry([c, f])))
        torch.cuda.synchronize()
        report("serial (staged)", 20, time.perf_counter() - t0)
    for sw in (0.005, 0.00005):
        sys.setswitchinterval(sw)
        for n in (2, 3, 4):
            pipe = BuildingPipeline(model, cfg, in_flight=n, device=dev)
            pipe.map([scenes[i % 4] for i in range(8)]); T.clear()
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pipe.map([scenes[i % 4] for i in range(40)])
                torch.cuda.synchronize()
                report(f"switch {sw} in flight {n}", 40, time.perf_counter() - t0)

"""Development probe: inference throughput of serving.BuildingPipeline for 1-4 buildings in flight vs the serial loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.serving import BuildingPipeline
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
scenes = [torch.from_numpy(make_scene(i, 500000)).to(dev) for i in range(4)]


def step(i):
    coords, feats = voxelize(scenes[i % 4], cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    return model([coords, feats])


with torch.no_grad():
    ref = [step(i) for i in range(4)]
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        print("serial: %.1f buildings/s" % (20 / (time.perf_counter() - t0)), flush=True)
    pipes = {n: BuildingPipeline(model, cfg, in_flight=n, device=dev) for n in (1, 2, 3, 4)}
    for n in (1, 2, 3, 4, 3, 2):
        pipe = pipes[n]
        pipe.map([scenes[i % 4] for i in range(8)])
        for total in (20, 20, 48, 48):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = pipe.map([scenes[i % 4] for i in range(total)])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            bad = sum(0 if all(o[k].shape == ref[i % 4][k].shape and torch.equal(o[k], ref[i % 4][k])
                               for k in ("bbox3d", "scores", "labels")) else 1 for i, o in enumerate(outs))
            print(f"in flight {n}: {total / dt:.1f} buildings/s  ({dt / total * 1e3:.2f} ms each), mismatching results {bad}",
                  flush=True)

"""Development probe: N host threads, each running whole bs=1 passes of its share of the buildings on a stream of
its own (no stage pipeline) -- against the serial loop and serving.BuildingPipeline."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
scenes = [torch.from_numpy(make_scene(i, 500000)).to(dev) for i in range(4)]
s = cfg.SPARSE3D


def step(i):
    coords, feats = voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
    return model([coords, feats])


with torch.no_grad():
    ref = [step(i) for i in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    print("serial: %.1f buildings/s" % (20 / (time.perf_counter() - t0)), flush=True)

for sw in (0.005, 0.0005):
    sys.setswitchinterval(sw)
    for n in (2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        for total in (8, 40, 40):
            outs = [None] * total
            errs = []

            def worker(t):
                try:
                    torch.cuda.set_device(dev)
                    with torch.no_grad(), torch.cuda.stream(streams[t]):
                        for i in range(t, total, n):
                            outs[i] = step(i)
                        streams[t].synchronize()
                except BaseException as e:
                    errs.append(e)

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=worker, args=(t,)) for t in range(n)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if errs:
                print("error:", repr(errs[0])[:300], flush=True)
                break
            bad = sum(0 if all(o[k].shape == ref[i % 4][k].shape and torch.equal(o[k], ref[i % 4][k])
                               for k in ("bbox3d", "scores", "labels")) else 1 for i, o in enumerate(outs))
            print(f"switch {sw} threads {n}: {total / dt:.1f} buildings/s ({dt / total * 1e3:.2f} ms each), mismatching {bad}", flush=True)

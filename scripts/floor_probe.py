"""Development probe: GPU time of the feature pass and of the tail alone (geometry built beforehand, device idle
otherwise) -- what a pass would take if grids and rulebooks cost nothing -- next to the whole pass."""
import os, sys, time
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
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
with torch.no_grad():
    acc = [0.0, 0.0, 0.0, 0.0]
    n = 0
    for i in range(24):
        c, f = voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
        torch.cuda.synchronize()
        ev[0].record()
        net = model.stage_geometry([c, f])
        ev[1].record()
        torch.cuda.synchronize()
        ev[2].record()
        feats = model.stage_features(net)
        ev[3].record()
        torch.cuda.synchronize()
        t_geo, t_feat = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
        ev[0].record()
        out = model.stage_tail(feats)
        ev[1].record()
        torch.cuda.synchronize()
        t_tail = ev[0].elapsed_time(ev[1])
        ev[0].record()
        model([scenes[i % 4]]) if False else None
        if i >= 4:
            acc[0] += t_geo; acc[1] += t_feat; acc[2] += t_tail; n += 1
    print(f"geometry alone {acc[0]/n:.3f} ms, feature pass alone {acc[1]/n:.3f} ms, tail alone {acc[2]/n:.3f} ms", flush=True)

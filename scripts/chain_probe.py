"""Grid chain vs one d3d_conv_prepare per level on the same input: every map the backbone hands on, bit for bit.
  python scripts/chain_probe.py [examples] [points per example]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from detection_3d_amd._lib import lib
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
cs, fs = [], []
for b in range(B):
    c, f = voxelize(torch.from_numpy(make_scene(40 + b, n, (35.0, 27.0, 2.7) if n > 600000 else (25.0, 19.0, 2.7))).to(dev), 50,
                    cfg.SPARSE3D.VOXEL_FULL_SCALE)
    cs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64, device=dev)], 1) if B > 1 else c)
    fs.append(f)
coords, feats = torch.cat(cs), torch.cat(fs)
outs = {}
with torch.no_grad():
    for on in (0, 1, 0, 1):
        lib().d3d_grid_chain_enable(on)
        rpn, roi = model.backbone([coords, feats])
        torch.cuda.synchronize()
        maps = [(t.features.clone(), t.get_spatial_locations().clone()) for t in rpn + roi]
        if on in outs:
            for (a, la), (b_, lb) in zip(outs[on], maps):
                print("repeat chain=%d identical:" % on, torch.equal(a, b_), torch.equal(la, lb))
        outs[on] = maps
for i, ((a, la), (b_, lb)) in enumerate(zip(outs[0], outs[1])):
    same_l = la.shape == lb.shape and torch.equal(la, lb)
    same_f = a.shape == b_.shape and torch.equal(a, b_)
    err = (a - b_).abs().max().item() / max(a.abs().max().item(), 1e-30) if a.shape == b_.shape else float("nan")
    print(f"map {i}: rows {tuple(a.shape)} vs {tuple(b_.shape)} locations equal {same_l} features equal {same_f} rel err {err:.3g}")

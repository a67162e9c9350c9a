"""Weight-sharing convolution kernel (conv_ws.hip) against k_conv on the same building: every map the backbone hands on bit for
bit, then the time of the sparse convolutions per family with either kernel.
  python scripts/conv_ws_probe.py [points]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detection_3d_amd._lib import lib
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.sparseconvnet import SCN
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
pcl = torch.from_numpy(make_scene(3, n)).to(dev)
outs = {}
with torch.no_grad():
    coords, feats = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    for mode in (0, 2, 1):
        lib().d3d_conv_ws_mode(mode)
        rpn, roi = model.backbone([coords, feats])
        torch.cuda.synchronize()
        outs[mode] = [t.features.clone() for t in rpn + roi]
    for mode in (2, 1):
        same = [torch.equal(a, b) for a, b in zip(outs[0], outs[mode])]
        err = [((a - b).abs().max() / a.abs().max()).item() for a, b in zip(outs[0], outs[mode])]
        print(f"mode {mode} vs k_conv: identical {same}  rel err {['%.2g' % e for e in err]}")
    lib().d3d_conv_late_mode(0)
    lib().d3d_conv_ws_mode(0)
    ref = [t.features.clone() for t in sum(model.backbone([coords, feats]), [])]
    lib().d3d_conv_late_mode(1)
    got = [t.features.clone() for t in sum(model.backbone([coords, feats]), [])]
    print("late gathers vs early: identical", [torch.equal(a, b) for a, b in zip(ref, got)])
    for mode, late in ((0, 0), (0, 1), (1, 1), (0, 0), (0, 1), (1, 1)):
        lib().d3d_conv_ws_mode(mode)
        lib().d3d_conv_late_mode(late)
        prof = SCN.ConvProfiler()
        SCN.set_profiler(prof)
        prof.start_scene("s", True)
        model.backbone([coords, feats])
        torch.cuda.synchronize()
        prof.records = []
        for _ in range(10):
            prof.start_scene("s", False)
            model.backbone([coords, feats])
        torch.cuda.synchronize()
        summ = prof.summary()
        SCN.set_profiler(None)
        tot = sum(v["ms"] for v in summ.values()) / 10
        parts = "  ".join(f"{k}: {v['ms'] / 10:.3f} ms {v['flops'] / (v['ms'] * 1e-3) / 1e12:.0f} TF"
                          for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:5])
        print(f"ws {mode} late {late}: all sparse convs {tot:.3f} ms per building | {parts}")

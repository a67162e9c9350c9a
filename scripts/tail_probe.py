"""Development probe: host / GPU timeline of the detector tail (RPN, pooler + box head, post-processing)."""
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
pcl = torch.from_numpy(make_scene(0, 500000)).to(dev)
LOG = []


def wrap(obj, name, tag):
    target = getattr(obj, name)
    if isinstance(target, torch.nn.Module):     # wrap the child's forward
        obj, name = target, "forward"
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        r = fn(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        LOG.append((tag, t0, time.perf_counter(), e0, e1))
        return r
    object.__setattr__(obj, name, w)


box = model.roi_heads.box
wrap(model, "backbone", "backbone")
wrap(model.rpn, "head", "rpn.head")
wrap(model.rpn, "anchor_generator", "rpn.anchors")
wrap(model.rpn, "select_proposals", "rpn.select")
wrap(model, "rpn", "rpn (all)")
wrap(box, "feature_extractor", "box.features")
wrap(box, "predictor", "box.predictor")
wrap(box, "post_processor", "box.post")
with torch.no_grad():
    for rep in range(5):
        LOG.clear()
        torch.cuda.synchronize()
        E0 = torch.cuda.Event(enable_timing=True); E0.record(); T0 = time.perf_counter()
        c, f = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        out = model([c, f])
        torch.cuda.synchronize()
        T1 = time.perf_counter()
    print(f"step {1e3 * (T1 - T0):.2f} ms")
    for tag, t0, t1, e0, e1 in LOG:
        print(f"  {tag:14s} host {1e3 * (t0 - T0):6.2f} -> {1e3 * (t1 - T0):6.2f}   gpu {E0.elapsed_time(e0):6.2f} -> {E0.elapsed_time(e1):6.2f}")

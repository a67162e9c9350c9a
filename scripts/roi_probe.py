"""Development probe: time of the pooler (d3d_roi_prepare + k_roi_sparse per level) on the bench scene's RoIs."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
with torch.no_grad():
    res, mid = model(list(voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)), return_intermediates=True)
    fe = model.roi_heads.box.feature_extractor
    x0, p = mid["roi_features"], mid["proposals"]
    for _ in range(5):
        out = fe.pooler.pool_metric(x0, p, fe.voxel_scale, channels_inner=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        out = fe.pooler.pool_metric(x0, p, fe.voxel_scale, channels_inner=True)
    torch.cuda.synchronize()
    print(f"pooler: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us for {p.shape[0]} RoIs, checksum {out.double().sum().item():.6f}")

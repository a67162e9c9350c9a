"""Development probe: training-step time of configs/6c fpn4321 bs=1 fp32 on the 500k-point synthetic scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene, make_targets
from detection_3d_amd.voxelize import voxelize
from detection_3d_amd import training as T

name = sys.argv[1] if len(sys.argv) > 1 else "6c_Fpn4321"
dev = torch.device("cuda:0")
cfg = get_cfg(name)
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).train()
opt = T.make_optimizer(cfg, model)
pcl = torch.from_numpy(make_scene(0, 500000)).to(dev)
boxes, labels = make_targets(0)
targets = {"bbox3d": torch.from_numpy(boxes).to(dev), "labels": torch.from_numpy(labels).to(dev)}
for it in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    coords, feats = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    losses = model([coords, feats], targets)
    total = sum(losses.values())
    torch.cuda.synchronize(); t1 = time.time()
    opt.zero_grad(); total.backward()
    torch.cuda.synchronize(); t2 = time.time()
    opt.step()
    torch.cuda.synchronize(); t3 = time.time()
    print(f"{name} it {it}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.1f} ms  loss {total.item():.4f}  "
          f"mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)

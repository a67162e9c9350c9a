import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.scene_io import ScenePrefetcher, load_scene, scene_targets
from detection_3d_amd.synthetic import write_scene_file
cfg = get_cfg("6c_Fpn4321")
d = tempfile.mkdtemp()
files = [write_scene_file(os.path.join(d, f"s{i}.npz"), i, 500000, cfg.INPUT.CLASSES) for i in range(4)]
dev = torch.device("cuda:0")
t0 = time.perf_counter(); pcl, boxes = load_scene(files[0]); t1 = time.perf_counter()
tg = scene_targets(pcl, boxes, cfg.INPUT.CLASSES, 50); t2 = time.perf_counter()
h = torch.from_numpy(pcl).pin_memory(); t3 = time.perf_counter()
x = h.to(dev, non_blocking=True); torch.cuda.synchronize(); t4 = time.perf_counter()
print(f"load {1e3*(t1-t0):.1f} ms targets {1e3*(t2-t1):.1f} pin {1e3*(t3-t2):.1f} h2d {1e3*(t4-t3):.1f}")
for rep in range(3):
    t0 = time.perf_counter(); n = 0
    for pcl, tg, p in ScenePrefetcher(files * 3, cfg.INPUT.CLASSES, 50, device=dev, depth=2):
        n += 1
    torch.cuda.synchronize()
    print(f"prefetcher alone: {1e3*(time.perf_counter()-t0)/n:.1f} ms per building")

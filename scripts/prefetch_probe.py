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
for w in (1, 2, 3, 6):
    t0 = time.perf_counter(); n = 0
    for pcl, tg, p in ScenePrefetcher(files * 4, cfg.INPUT.CLASSES, 50, device=dev, depth=2 * w, workers=w):
        n += 1
    torch.cuda.synchronize()
    print(f"workers {w}: {1e3*(time.perf_counter()-t0)/n:.1f} ms per building")
from detection_3d_amd import scene_io
buf = torch.empty(500000 * 9 + 1024, dtype=torch.float32).pin_memory()
for rep in range(3):
    t0 = time.perf_counter(); v, b = scene_io.load_scene_into(files[0], lambda n: buf.numpy()); t1 = time.perf_counter()
    tg = scene_targets(v, b, cfg.INPUT.CLASSES, 50); t2 = time.perf_counter()
    x = torch.from_numpy(v).to(dev, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"read-into {1e3*(t1-t0):.2f} ms targets {1e3*(t2-t1):.2f} h2d {1e3*(t3-t2):.2f}")
# the whole inference loop from files (engine.inference) against the same buildings resident in HBM
from detection_3d_amd import engine
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.voxelize import voxelize
cfg4 = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg4).to(dev).eval()
many = files * 6
engine.inference(model, cfg4, many[:4], dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
dets, gts = engine.inference(model, cfg4, many, dev)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"engine.inference from files: {len(many) / dt:.1f} buildings/s ({1e3 * dt / len(many):.2f} ms per building)")
clouds = [torch.from_numpy(load_scene(f)[0]).to(dev) for f in files]
with torch.no_grad():
    for i in range(4):
        model(list(voxelize(clouds[i % 4], 50, cfg4.SPARSE3D.VOXEL_FULL_SCALE)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(many)):
        model(list(voxelize(clouds[i % 4], 50, cfg4.SPARSE3D.VOXEL_FULL_SCALE)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"resident clouds:             {len(many) / dt:.1f} buildings/s ({1e3 * dt / len(many):.2f} ms per building)")

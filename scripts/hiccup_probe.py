"""Development probe: per-pass host times of 300 serial passes -- how often a pass takes much longer than the median, with
the garbage collector as it is, frozen after warm-up (gc.freeze) or disabled."""
import gc, os, sys, time
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def run(tag):
    ts = []
    with torch.no_grad():
        for i in range(10):
            model(list(voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)))
        torch.cuda.synchronize()
        t = time.perf_counter()
        allocs = []
        for i in range(N):
            a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
            model(list(voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)))
            t2 = time.perf_counter()
            ts.append(1e3 * (t2 - t))
            allocs.append(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0)
            t = time.perf_counter()
    a = sorted(ts)
    slow = [round(x, 1) for x in ts if x > 1.25 * a[len(a) // 2]]
    med = a[len(a) // 2]
    print(f"   device allocations in the timed passes: {sum(allocs)}; in the slow passes: "
          f"{[(round(x, 1), n) for x, n in zip(ts, allocs) if x > 1.25 * med]}", flush=True)
    print(f"{tag}: mean {sum(ts) / len(ts):.3f} median {a[len(a) // 2]:.3f} p99 {a[int(0.99 * len(a))]:.3f} max {a[-1]:.3f}; "
          f"{len(slow)} passes over 1.25 x median: {slow[:12]}; gc counts {gc.get_count()}", flush=True)


run("gc default")
run("gc default")
gc.collect()
gc.freeze()
run("gc.freeze after warm-up")
run("gc.freeze after warm-up")
gc.disable()
run("gc disabled")
run("gc disabled")

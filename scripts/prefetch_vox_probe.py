"""Development probe: the next building's voxelization enqueued by a helper thread on a side stream while the detector
runs the current one (what the reference's DataLoader workers do on the CPU) vs the serial loop."""
import os, sys, time, threading, queue
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
N = 40


def serial():
    for i in range(N):
        c, f = voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
        model([c, f])


side = torch.cuda.Stream(device=dev)


def prefetched():
    q = queue.Queue(maxsize=1)
    main = torch.cuda.current_stream(dev)

    def worker():
        torch.cuda.set_device(dev)
        with torch.no_grad(), torch.cuda.stream(side):
            for i in range(N):
                c, f = voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
                ev = torch.cuda.Event()
                ev.record(side)
                q.put((c, f, ev))
    th = threading.Thread(target=worker)
    th.start()
    for i in range(N):
        c, f, ev = q.get()
        main.wait_event(ev)
        c.record_stream(main); f.record_stream(main)
        model([c, f])
    th.join()


with torch.no_grad():
    for name, fn in (("serial", serial), ("prefetched", prefetched), ("serial", serial), ("prefetched", prefetched)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per building", flush=True)

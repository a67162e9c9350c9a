"""Development probe: host / GPU timeline of the two-lane backbone pass (per level: when the geometry step returns on
the host, when the level's feature kernels finish on the GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.sparseconvnet import fpn_net
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
pcl = torch.from_numpy(make_scene(0, 500000)).to(dev)
bb = model.backbone
LOG = []
orig_steps = bb._geometry_steps
orig_run = bb._run_down


def steps(net, full):
    g = orig_steps(net, full)
    while True:
        t0 = time.perf_counter()
        try:
            k = next(g)
        except StopIteration:
            return
        LOG.append(("G%d host" % k, t0, time.perf_counter()))
        yield k


def run_down(m, net):
    t0 = time.perf_counter()
    out = orig_run(m, net)
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    LOG.append(("F host", t0, time.perf_counter(), ev))
    return out


bb._geometry_steps = steps
bb._run_down = run_down
with torch.no_grad():
    for two in (True, False, True, False):
        fpn_net.TWO_LANE = two
        for rep in range(4):
            LOG.clear()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e0.record(); T0 = time.perf_counter()
            c, f = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
            out = bb([c, f])
            t_enq = time.perf_counter()
            torch.cuda.synchronize()
            t_end = time.perf_counter()
        print(f"two_lane={two}: backbone enqueued at {1e3 * (t_enq - T0):.2f} ms, GPU done at {1e3 * (t_end - T0):.2f} ms")
        for rec in LOG:
            if len(rec) == 3:
                print(f"   {rec[0]:8s} host {1e3 * (rec[1] - T0):6.2f} -> {1e3 * (rec[2] - T0):6.2f}")
            else:
                print(f"   {rec[0]:8s} host {1e3 * (rec[1] - T0):6.2f} -> {1e3 * (rec[2] - T0):6.2f}   gpu done {e0.elapsed_time(rec[3]):6.2f}")

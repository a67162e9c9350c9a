"""Development probe: level-0 3x3x3 rulebook build (hash probes + sort + transpose) for input points in random order
vs sorted along a Morton curve (spatially coherent site ids): how much of the probe time is locality."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize
from detection_3d_amd import sparseconvnet as scn
from detection_3d_amd.sparseconvnet import SCN

dev = torch.device("cuda:0")


def morton(c):
    c = c.astype(np.uint64)
    key = np.zeros(c.shape[0], np.uint64)
    for b in range(12):
        for d in range(3):
            key |= ((c[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return key


for n_pts, batch in ((500000, 1), (1000000, 4)):
    for order in ("random", "morton"):
        cs, fs = [], []
        for b in range(batch):
            pcl = torch.from_numpy(make_scene(b, n_pts)).to(dev)
            c, f = voxelize(pcl, 50, [4096, 4096, 512])
            c, f = c.cpu().numpy(), f.cpu().numpy()
            if order == "morton":
                o = np.argsort(morton(c[:, :3]), kind="stable")
                c, f = c[o], f[o]
            cc = np.concatenate([c[:, :3], np.full((c.shape[0], 1), b, c.dtype)], 1)
            cs.append(cc); fs.append(f)
        coords = torch.from_numpy(np.concatenate(cs)).to(dev)
        feats = torch.from_numpy(np.concatenate(fs)).to(dev)
        ts = []
        with torch.no_grad():
            for rep in range(4):
                layer = scn.InputLayer(3, [4096, 4096, 512], mode=4)
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                torch.cuda.synchronize()
                e[0].record()
                t = layer([coords, feats, batch])
                e[1].record()
                SCN.SubmanifoldConvolution_prepare(t.spatial_size, (3, 3, 3), t.metadata)
                e[2].record()
                torch.cuda.synchronize()
                ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), t.features.shape[0]))
                del t, layer
        print(f"{batch} x {n_pts} points, {order}: input layer {ts[-1][0]:.3f} ms, 3x3x3 rulebook {ts[-1][1]:.3f} ms, {ts[-1][2]} sites", flush=True)

"""Development probe: how many DISTINCT offset masks the rows of each level's 3x3x3 plan have, how long the runs of equal
block masks are, and what sharing a weight tile among R consecutive row blocks would cost in executed steps (union of
their masks) -- the numbers behind DESIGN 5c / 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from detection_3d_amd import sparseconvnet as scn
from detection_3d_amd.sparseconvnet import SCN
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
pcl = torch.from_numpy(make_scene(0, 500000)).to(dev)
size = [4096, 4096, 512]
c, f = voxelize(pcl, 50, size)
t = scn.InputLayer(3, size, mode=4)([c, f])
m = t.metadata
cur = list(size)
for k in range(5):
    SCN.SubmanifoldConvolution_prepare(tuple(cur), (3, 3, 3), m)
    n = m.getNActive(tuple(cur))
    trip = m.export_rules(0, tuple(cur), (3, 3, 3)).cpu().numpy().astype(np.int64)     # (in, out, offset)
    mask = np.zeros(n, np.int64)
    np.bitwise_or.at(mask, trip[:, 1], 1 << trip[:, 2])
    pop = np.array([bin(v).count("1") for v in np.unique(mask)])
    order = np.argsort(-mask, kind="stable")                                   # the plan's order: by mask, descending
    sm = mask[order]
    nb = (n + 31) // 32
    pad = np.zeros(nb * 32, np.int64); pad[:n] = sm
    blk = np.bitwise_or.reduce(pad.reshape(nb, 32), axis=1)
    popc = lambda a: np.array([bin(int(v)).count("1") for v in a])
    useful = popc(sm).sum()
    line = f"scale {k}: rows {n} blocks {nb} distinct row masks {len(np.unique(mask))} distinct block masks {len(np.unique(blk))}"
    for R in (1, 2, 4, 8):
        g = (nb + R - 1) // R
        bp = np.zeros(g * R, np.int64); bp[:nb] = blk
        uni = np.bitwise_or.reduce(bp.reshape(g, R), axis=1)
        executed = (popc(uni) * R * 32).sum()
        line += f"  R={R}: executed/useful {executed / useful:.3f}"
    print(line, flush=True)
    nxt = [v // 2 for v in cur]
    SCN.Convolution_prepare(tuple(cur), tuple(nxt), (2, 2, 2), (2, 2, 2), m)
    cur = nxt

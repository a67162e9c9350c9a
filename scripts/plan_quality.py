"""Development probe: grouping quality (executed / useful row-offset steps) and build time of the submanifold 3x3x3 plans of
the bench building, per scale:  [D3D_PLAN_GROUP=0] python scripts/plan_quality.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd._lib import check, ints, lib, stream_of
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
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    SCN.SubmanifoldConvolution_prepare(tuple(cur), (3, 3, 3), m)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    nb, ex, nr = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0)
    check(lib().d3d_plan_stats(m._h, 0, ints(cur), ints([3, 3, 3]), ints([0, 0, 0]), ctypes.byref(nb), ctypes.byref(ex),
                               ctypes.byref(nr), stream_of()))
    print(f"scale {k}: rows {m.getNActive(tuple(cur))} blocks {nb.value} rules {nr.value} executed/useful {ex.value / max(nr.value, 1):.3f} "
          f"build {1e3 * (t1 - t0):.3f} ms")
    nxt = [v // 2 for v in cur]
    SCN.Convolution_prepare(tuple(cur), tuple(nxt), (2, 2, 2), (2, 2, 2), m)
    nb2, ex2, nr2 = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0)
    check(lib().d3d_plan_stats(m._h, 1, ints(cur), ints([2, 2, 2]), ints([2, 2, 2]), ctypes.byref(nb2), ctypes.byref(ex2),
                               ctypes.byref(nr2), stream_of()))
    print(f"   strided -> {nxt}: blocks {nb2.value} rules {nr2.value} executed/useful {ex2.value / max(nr2.value, 1):.3f}")
    cur = nxt

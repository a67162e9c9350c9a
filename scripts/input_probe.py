"""Development probe: the input layer alone (grid build, point lists, feature pass) at 500 k / 1 M points and a batch of 4."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detection_3d_amd import sparseconvnet as scn
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize
dev = torch.device("cuda:0")
size = [4096, 4096, 512]
def case(tag, n, B, ext):
    cs, fs = [], []
    for b in range(B):
        c, f = voxelize(torch.from_numpy(make_scene(b, n, ext)).to(dev), 50, size)
        cs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64, device=dev)], 1) if B > 1 else c)
        fs.append(f)
    c, f = torch.cat(cs), torch.cat(fs)
    layer = scn.InputLayer(3, size, mode=4)
    with torch.no_grad():
        for _ in range(3):
            t = layer([c, f] if B == 1 else [c, f, B])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            t = layer([c, f] if B == 1 else [c, f, B])
        torch.cuda.synchronize()
    print(f"{tag}: {c.shape[0]} points -> {t.features.shape[0]} sites, input layer {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
case("500k", 500000, 1, (25.0, 19.0, 2.7))
case("1M", 1000000, 1, (35.0, 27.0, 2.7))
case("4x1M", 1000000, 4, (35.0, 27.0, 2.7))

"""a1. Point cloud -> voxel coordinates on the GPU, arithmetic of
data3d/suncg_utils/suncg_dataset.py:97-177 (fp64 scale + per-axis min shift, bounds filter, trunc)."""
import ctypes

import torch

from ._lib import check, ints, lib, ptr, require_gpu, stream_of


def voxelize(pcl, scale=50, full_scale=(4096, 4096, 512)):
    """pcl fp32 [N, F] on the GPU (xyz first) -> (coords int64 [M,3], feats fp32 [M,F]); input order
    is preserved and points outside [0, full_scale) after the min-shift are dropped."""
    pcl = pcl.detach().to(torch.float32).contiguous()
    require_gpu(pcl)
    n, nfeat = pcl.shape
    coords = torch.empty((n, 3), dtype=torch.int64, device=pcl.device)
    feats = torch.empty((n, nfeat), dtype=torch.float32, device=pcl.device)
    nbytes = lib().d3d_voxelize_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pcl.device)
    kept = ctypes.c_int(0)
    check(lib().d3d_voxelize(ptr(pcl), n, nfeat, float(scale), ints(full_scale), ptr(coords), ptr(feats),
                             ctypes.byref(kept), ptr(scratch), nbytes, stream_of()))
    return coords[:kept.value], feats[:kept.value]

"""maskrcnn_benchmark/layers/roi_align_rotated_3d.py mirror: `_C.roi_align_rotated_3d_forward`
(csrc/ROIAlignRotated3D.h:10-26) on the MI355X, plus a sparse-input variant that samples the
SparseConvNetTensor through its hash grid (no dense [1,128,256,256,32] = 1.07 GB map)."""
import torch

from ._lib import check, ints, lib, ptr, require_gpu, stream_of


def roi_align_rotated_3d_forward(input, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                 sampling_ratio):
    """input [B,C,H,W,Z] fp32, rois [K,8] = (batch, cw, ch, cz, w, h, z, theta_deg) -> [K,C,ph,pw,pz]."""
    inp = input.detach().to(torch.float32).contiguous()
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(inp, r)
    B, C, H, W, Z = inp.shape
    K = r.shape[0]
    out = torch.empty((K, C, pooled_height, pooled_width, pooled_zsize), dtype=torch.float32, device=inp.device)
    check(lib().d3d_roi_align_rotated_3d_forward(ptr(inp), B, C, H, W, Z, ptr(r), K, float(spatial_scale),
                                                 pooled_height, pooled_width, pooled_zsize,
                                                 int(sampling_ratio), ptr(out), stream_of()))
    return out


def roi_align_rotated_3d_sparse(feat_s3d, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                sampling_ratio, crop=None):
    """Equals roi_align_rotated_3d_forward(sparse_3d_to_dense_2d(feat_s3d), ...)
    (sparseconvnet/tools_3d_2d.py:7-48 crops the dense map to the occupied extent = `crop`)."""
    f = feat_s3d.features.contiguous()
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(f, r)
    if crop is None:
        loc = feat_s3d.get_spatial_locations()
        crop = (loc[:, :3].max(0)[0] + 1).tolist()
    K, C = r.shape[0], f.shape[1]
    out = torch.empty((K, C, pooled_height, pooled_width, pooled_zsize), dtype=torch.float32, device=f.device)
    check(lib().d3d_roi_align_rotated_3d_sparse_forward(
        feat_s3d.metadata._h, ints(feat_s3d.spatial_size.tolist()), ptr(f), C, ints(crop), ptr(r), K,
        float(spatial_scale), pooled_height, pooled_width, pooled_zsize, int(sampling_ratio), ptr(out),
        stream_of()))
    return out


class ROIAlignRotated3D(torch.nn.Module):
    """maskrcnn_benchmark/layers/roi_align_rotated_3d.py:56-75 (forward only)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    @torch.no_grad()
    def forward(self, input, rois):
        ph, pw, pz = self.output_size
        if hasattr(input, "metadata"):
            return roi_align_rotated_3d_sparse(input, rois, self.spatial_scale, ph, pw, pz, self.sampling_ratio)
        return roi_align_rotated_3d_forward(input, rois, self.spatial_scale, ph, pw, pz, self.sampling_ratio)

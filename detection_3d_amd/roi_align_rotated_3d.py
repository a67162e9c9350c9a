"""maskrcnn_benchmark/layers/roi_align_rotated_3d.py mirror: `_C.roi_align_rotated_3d_forward`
(csrc/ROIAlignRotated3D.h:10-26) on the MI355X, plus a sparse-input variant that samples the
SparseConvNetTensor through its hash grid (no dense [1,128,256,256,32] = 1.07 GB map)."""
import torch

from ._lib import check, floats, ints, lib, ptr, require_gpu, stream_of


def roi_align_rotated_3d_backward(grad, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                  batch_size, channels, height, width, zsize, sampling_ratio):
    """_C.roi_align_rotated_3d_backward (csrc/ROIAlignRotated3D.h:28-47) -> dense gradient [B,C,H,W,Z]."""
    g = grad.detach().to(torch.float32).contiguous()
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(g, r)
    out = torch.empty((batch_size, channels, height, width, zsize), dtype=torch.float32, device=g.device)
    check(lib().d3d_roi_align_rotated_3d_backward(ptr(g), batch_size, channels, height, width, zsize, ptr(r),
                                                  r.shape[0], float(spatial_scale), pooled_height, pooled_width,
                                                  pooled_zsize, int(sampling_ratio), ptr(out), stream_of()))
    return out


class _RoiDenseFn(torch.autograd.Function):
    """layers/roi_align_rotated_3d.py:11-51 (_ROIAlignRotated3D)."""

    @staticmethod
    def forward(ctx, inp, rois, spatial_scale, ph, pw, pz, sampling_ratio):
        B, C, H, W, Z = inp.shape
        K = rois.shape[0]
        out = torch.empty((K, C, ph, pw, pz), dtype=torch.float32, device=inp.device)
        check(lib().d3d_roi_align_rotated_3d_forward(ptr(inp), B, C, H, W, Z, ptr(rois), K, float(spatial_scale),
                                                     ph, pw, pz, int(sampling_ratio), ptr(out), stream_of()))
        ctx.save_for_backward(rois)
        ctx.args = (spatial_scale, ph, pw, pz, sampling_ratio, tuple(inp.shape))
        return out

    @staticmethod
    def backward(ctx, grad):
        (rois,) = ctx.saved_tensors
        spatial_scale, ph, pw, pz, sampling_ratio, shape = ctx.args
        d = roi_align_rotated_3d_backward(grad, rois, spatial_scale, ph, pw, pz, *shape, sampling_ratio)
        return d, None, None, None, None, None, None


def roi_align_rotated_3d_forward(input, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                 sampling_ratio):
    """input [B,C,H,W,Z] fp32, rois [K,8] = (batch, cw, ch, cz, w, h, z, theta_deg) -> [K,C,ph,pw,pz]."""
    inp = input.to(torch.float32).contiguous()
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(inp, r)
    return _RoiDenseFn.apply(inp, r, spatial_scale, pooled_height, pooled_width, pooled_zsize, sampling_ratio)


class _RoiSparseFn(torch.autograd.Function):
    """forward: d3d_roi_align_rotated_3d_sparse_forward; backward: the dense backward of
    layers/roi_align_rotated_3d.py:29-51 restricted to the active sites (scatter-add into feature rows)."""

    @staticmethod
    def forward(ctx, feats, rois, metadata, spatial_size, crop, spatial_scale, ph, pw, pz, sampling_ratio):
        f = feats.contiguous()
        K, C = rois.shape[0], f.shape[1]
        out = torch.empty((K, C, ph, pw, pz), dtype=torch.float32, device=f.device)
        check(lib().d3d_roi_align_rotated_3d_sparse_forward(
            metadata._h, ints(spatial_size), ptr(f), C, ints(crop), ptr(rois), K, float(spatial_scale), ph, pw, pz,
            int(sampling_ratio), None, 0, 0, ptr(out), stream_of()))
        ctx.save_for_backward(rois)
        ctx.args = (metadata, spatial_size, crop, spatial_scale, ph, pw, pz, sampling_ratio, f.shape)
        return out

    @staticmethod
    def backward(ctx, grad):
        (rois,) = ctx.saved_tensors
        metadata, spatial_size, crop, spatial_scale, ph, pw, pz, sampling_ratio, shape = ctx.args
        g = grad.contiguous()
        d_feats = torch.zeros(shape, dtype=torch.float32, device=g.device)
        check(lib().d3d_roi_align_rotated_3d_sparse_backward(
            metadata._h, ints(spatial_size), ptr(g), shape[1], ints(crop), ptr(rois), rois.shape[0],
            float(spatial_scale), ph, pw, pz, int(sampling_ratio), ptr(d_feats), stream_of()))
        return d_feats, None, None, None, None, None, None, None, None, None


def roi_align_rotated_3d_sparse(feat_s3d, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                sampling_ratio, crop=None):
    """Equals roi_align_rotated_3d_forward(sparse_3d_to_dense_2d(feat_s3d), ...)
    (sparseconvnet/tools_3d_2d.py:7-48 crops the dense map to the occupied extent = `crop`)."""
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(feat_s3d.features, r)
    if crop is None:
        loc = feat_s3d.get_spatial_locations()
        crop = (loc[:, :3].max(0)[0] + 1).tolist()
    return _RoiSparseFn.apply(feat_s3d.features, r, feat_s3d.metadata, feat_s3d.spatial_size.tolist(),
                              [int(c) for c in crop], spatial_scale, pooled_height, pooled_width, pooled_zsize,
                              sampling_ratio)


def roi_align_rotated_3d_sparse_into(out, feat_s3d, rois, spatial_scale, sampling_ratio, crop=None,
                                     roi_levels=None, level=0, channels_inner=True):
    """Inference-only form of the above for the multi-level pooler (poolers_3d.py:150-168): pools into `out`
    ([K, ph, pw, C, pz] if channels_inner else [K, C, ph, pw, pz]) the RoIs whose roi_levels[i] == level
    (all of them when roi_levels is None) and leaves the other slots untouched."""
    f = feat_s3d.features.contiguous()
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(f, r, out)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == r.shape[0]
    if channels_inner:
        _, ph, pw, C, pz = out.shape
    else:
        _, C, ph, pw, pz = out.shape
    assert C == f.shape[1]
    if roi_levels is not None:
        assert roi_levels.dtype == torch.int32 and roi_levels.is_contiguous() and roi_levels.shape[0] == r.shape[0]
    # crop None: the library finds the occupied extent of the map on the device (no host read-back)
    check(lib().d3d_roi_align_rotated_3d_sparse_forward(
        feat_s3d.metadata._h, ints(feat_s3d.spatial_size.tolist()), ptr(f), C,
        None if crop is None else ints([int(c) for c in crop]), ptr(r),
        r.shape[0], float(spatial_scale), ph, pw, pz, int(sampling_ratio), ptr(roi_levels), int(level),
        1 if channels_inner else 0, ptr(out), stream_of()))
    return out


def roi_align_rotated_3d_sparse_levels_into(out, maps, rois, scales, sampling_ratio, roi_levels, channels_inner=True):
    """All levels of the multi-level pooler in ONE launch (d3d_roi_align_rotated_3d_sparse_forward_levels): RoI i is
    pooled from maps[roi_levels[i]] into out[i]; rows with another level value (-1: padding) are left untouched."""
    import ctypes
    fs = [m.features.contiguous() for m in maps]
    r = rois.detach().to(torch.float32).contiguous()
    require_gpu(r, out, *fs)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == r.shape[0]
    if channels_inner:
        _, ph, pw, C, pz = out.shape
    else:
        _, C, ph, pw, pz = out.shape
    assert all(f.shape[1] == C and f.dtype == torch.float32 for f in fs) and 1 <= len(fs) <= 4
    assert all(m.metadata is maps[0].metadata for m in maps)
    if roi_levels is not None:
        assert roi_levels.dtype == torch.int32 and roi_levels.is_contiguous() and roi_levels.shape[0] == r.shape[0]
    sizes = ints(tuple(int(v) for m in maps for v in m.spatial_size.tolist()))
    check(lib().d3d_roi_align_rotated_3d_sparse_forward_levels(
        maps[0].metadata._h, len(fs), sizes, (ctypes.c_void_p * len(fs))(*[f.data_ptr() for f in fs]), C,
        floats(scales), ptr(r), r.shape[0], ph, pw, pz, int(sampling_ratio), ptr(roi_levels),
        1 if channels_inner else 0, ptr(out), stream_of()))
    return out


def roi_prepare(boxes_metric, voxel_scale, scales, canonical_size, batch_ids=None, count=None):
    """d3d_roi_prepare: metric yx_zb proposals [K,7] -> (rois [K,8] in pixels, levels int32 [K] or None for a single
    level) in one launch; equals convert_metric_to_pixel + convert_to_roi_format + Pooler.map_levels bit for bit.
    batch_ids: int32 [K] example index of every proposal (None: one example).
    count: int32 [1] on the device = how many of the K rows are real (d3d_roi_prepare_counted: the others get level -1
    and are pooled by no level); needs more than one level."""
    b = boxes_metric.detach().to(torch.float32).contiguous()
    require_gpu(b)
    K = b.shape[0]
    if batch_ids is not None:
        assert batch_ids.dtype == torch.int32 and batch_ids.is_contiguous() and batch_ids.shape[0] == K
    rois = torch.empty((K, 8), dtype=torch.float32, device=b.device)
    levels = torch.empty((K,), dtype=torch.int32, device=b.device) if len(scales) > 1 else None
    if count is not None:
        assert count.dtype == torch.int32 and count.is_cuda and levels is not None
        check(lib().d3d_roi_prepare_counted(ptr(b), K, ptr(count), float(voxel_scale), floats(scales), len(scales),
                                            float(canonical_size), ptr(batch_ids), ptr(rois), ptr(levels), stream_of()))
        return rois, levels
    check(lib().d3d_roi_prepare(ptr(b), K, float(voxel_scale), floats(scales), len(scales), float(canonical_size),
                                ptr(batch_ids), ptr(rois), ptr(levels), stream_of()))
    return rois, levels


class ROIAlignRotated3D(torch.nn.Module):
    """maskrcnn_benchmark/layers/roi_align_rotated_3d.py:56-75 (forward only)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        ph, pw, pz = self.output_size
        if hasattr(input, "metadata"):
            return roi_align_rotated_3d_sparse(input, rois, self.spatial_scale, ph, pw, pz, self.sampling_ratio)
        return roi_align_rotated_3d_forward(input, rois, self.spatial_scale, ph, pw, pz, self.sampling_ratio)

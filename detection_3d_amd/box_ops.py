"""Rotated IoU / NMS / box codec ops with the reference's Python-level names and argument meaning,
running entirely on the GPU through libd3d_hip.so (no numpy / numba / spconv round trips).

    rotate_iou_gpu_eval   second/core/non_max_suppression/nms_gpu.py:614-650
    boxes_iou_3d          utils3d/rotate_nms_3d_torch.py:23-88
    rotate_nms_3d         second/pytorch/core/box_torch_ops.py:489-514
    nms_3d_clamped        maskrcnn_benchmark/structures/boxlist_ops_3d.py:14-62 (tensor form)
    box_decode            maskrcnn_benchmark/modeling/box_coder_3d.py:38-65
    limit_period          utils3d/geometric_torch.py:4-10
"""
import math

import torch

from . import _lib
from ._lib import D3DError, check, floats, lib, ptr, require_gpu, stream_of


def limit_period(val, offset, period):
    return val - torch.floor(val / period + offset) * period


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5] (xc, yc, d0, d1, angle) -> iou [N,K] (torch, on the GPU).
    Includes check_same_boxes (nms_gpu.py:653-664)."""
    boxes, query_boxes = _f32c(boxes), _f32c(query_boxes)
    require_gpu(boxes, query_boxes)
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = torch.zeros((n, k), dtype=torch.float32, device=boxes.device)
    if n and k:
        check(lib().d3d_rotate_iou_eval(ptr(boxes), n, ptr(query_boxes), k, int(criterion), ptr(out),
                                        stream_of()))
    return out


_FLAG_RULES = {
    # flag -> checker(aug) reproducing the asserts of rotate_nms_3d_torch.py:32-48
    "rpn_label_generation": lambda a: a["anchor_Y"] == 0 and a["target_Y"] >= 0.3,
    "roi_label_generation": lambda a: a["anchor_Y"] >= 0.3 and a["target_Y"] >= 0.3,
    "eval": lambda a: a["anchor_Y"] <= 0.3 and a["target_Y"] <= 0.3,
}


def boxes_iou_3d(targets_bbox3d, anchors_bbox3d, aug_thickness=None, criterion=-1, only_xy=False, flag=''):
    """targets [M,7], anchors [N,7] in yx_zb mode -> iou3d [M,N] = BEV rotated IoU x z-interval IoU."""
    if flag in _FLAG_RULES:
        assert _FLAG_RULES[flag](aug_thickness), (flag, aug_thickness)
    elif flag in ("rpn_post", "roi_post"):
        assert aug_thickness is None
    else:
        raise NotImplementedError(f"boxes_iou_3d: unknown flag {flag!r}")
    if aug_thickness is None:
        aug_thickness = {'target_Y': 0.0, 'target_Z': 0.0, 'anchor_Y': 0.0, 'anchor_Z': 0.0}
    t, a = _f32c(targets_bbox3d), _f32c(anchors_bbox3d)
    require_gpu(t, a)
    m, n = t.shape[0], a.shape[0]
    out = torch.zeros((m, n), dtype=torch.float32, device=t.device)
    if m and n:
        aug = floats([aug_thickness['target_Y'], aug_thickness['target_Z'], aug_thickness['anchor_Y'],
                      aug_thickness['anchor_Z']])
        check(lib().d3d_boxes_iou_3d(ptr(t), m, ptr(a), n, aug, int(criterion), int(bool(only_xy)), ptr(out),
                                     stream_of()))
    return out


_NMS_SCRATCH = {}


def _nms_key(dev):      # one scratch per (device, stream): buildings in flight on different streams never share it
    return (dev.index, _lib.raw_stream(dev))


def _nms_sorted(boxes_sorted, thresh, max_keep=0):
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    nbytes = lib().d3d_nms_scratch_bytes(n)
    buf = _NMS_SCRATCH.get(_nms_key(dev))
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _NMS_SCRATCH[_nms_key(dev)] = buf
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().d3d_rotate_nms_3d_sorted(ptr(boxes_sorted), n, float(thresh), int(max_keep or 0), ptr(keep), ptr(nk),
                                         ptr(buf), buf.numel(), stream_of()))
    return keep, nk


def nms_3d_batched(boxes, order, counts, n_max, iou_threshold, aug_thickness=(0.0, 0.0), max_keep=0, segments=1):
    """d3d_rotate_nms_3d_batched: `boxes` [M,7]; segment b's candidates are boxes[order[b, i]], i < counts[b]
    (order [B, stride] int32 in descending score order; None: `segments` lists laid out one after the other, segment b =
    boxes[b * n_max : (b + 1) * n_max] as they are; counts [B] int32 on the device or None) -> keep int32 [B, n_max]
    (box indices, selection order), n_keep [B]."""
    boxes = _f32c(boxes)
    require_gpu(boxes)
    dev = boxes.device
    if order is None:
        B, stride = int(segments), int(n_max)
        assert boxes.shape[0] >= B * n_max
    else:
        assert order.dtype == torch.int32 and order.is_contiguous() and order.dim() == 2
        B, stride = order.shape
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.is_contiguous() and counts.shape[0] == B
    nbytes = lib().d3d_nms_batched_scratch_bytes(B, n_max)
    buf = _NMS_SCRATCH.get(_nms_key(dev))
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _NMS_SCRATCH[_nms_key(dev)] = buf
    keep = torch.empty((B, max(n_max, 1)), dtype=torch.int32, device=dev)
    nk = torch.empty(B, dtype=torch.int32, device=dev)      # every segment's count is written (zeros for n_max == 0)
    check(lib().d3d_rotate_nms_3d_batched(ptr(boxes), ptr(order), stride, ptr(counts), B, n_max, float(iou_threshold),
                                          float(aug_thickness[0]), float(aug_thickness[1]), int(max_keep or 0),
                                          ptr(keep), ptr(nk), ptr(buf), buf.numel(), stream_of()))
    return keep, nk


def nms_3d_presorted(bbox3d, nms_thresh, nms_aug_thickness=None, max_proposals=-1, flag=''):
    """nms_3d_clamped for boxes already in descending score order (the RPN's top-k): no re-sort, the size clamp
    happens inside the kernel.  -> kept positions (int64), in score order."""
    if nms_aug_thickness is None:
        nms_aug_thickness = [0, 0]
    if flag == 'rpn_post':
        assert max_proposals > 100, max_proposals
    elif flag == 'roi_post':
        assert max_proposals == -1
    else:
        raise NotImplementedError(flag)
    if max_proposals < 0:
        max_proposals = 500
    n = min(bbox3d.shape[0], 2000)                                       # pre_max_size
    if n == 0:
        return torch.zeros([0], dtype=torch.int64, device=bbox3d.device)
    keep, nk = nms_3d_batched(bbox3d, None, None, n, nms_thresh, nms_aug_thickness, max_proposals)
    return keep[0, :int(nk.item())].long()


def topk_max():
    return int(lib().d3d_topk_max())


def topk_segments(vals, k, n=None, elem_stride=1, group_stride=0, n_groups=1, example=None, n_examples=1, sigmoid=False,
                  min_value=None, idx_map=None, reg=None, anchors=None, clip=10000.0, want_idx64=True, want_idx32=False):
    """d3d_topk_segments: the k best elements of every (example, group) segment of `vals` in ONE launch -- descending,
    equal scores: lower index first (the order the oracle port defines; torch.topk leaves it open).  Element i of group
    g is vals.view(-1)[g * group_stride + i * elem_stride]; `example` int32 [n] picks the elements of each example.
    reg [n, >= 7 groups] + anchors [n, 7]: also decode the selected rows (BoxCoder3D.decode, unit weights).
    -> dict(idx (int64 [S, k]) / idx32, scores [S, k], props [S, k, 7] or None, counts int32 [S]); rows past
    counts-of-elements are uninitialised.  min_value: counts = kept elements with value > min_value."""
    require_gpu(vals)
    assert vals.dtype == torch.float32 and vals.is_contiguous()
    dev = vals.device
    if n is None:
        n = vals.shape[0]
    S = n_groups * n_examples
    k = int(k)
    out = {"idx": torch.empty((S, max(k, 1)), dtype=torch.int64, device=dev) if want_idx64 else None,
           "idx32": torch.empty((S, max(k, 1)), dtype=torch.int32, device=dev) if want_idx32 else None,
           "scores": torch.empty((S, max(k, 1)), dtype=torch.float32, device=dev),
           "props": torch.empty((S, max(k, 1), 7), dtype=torch.float32, device=dev) if reg is not None else None,
           "counts": torch.empty((S,), dtype=torch.int32, device=dev)}
    if example is not None:
        assert example.dtype == torch.int32 and example.is_contiguous() and example.shape[0] == n
    if reg is not None:
        require_gpu(reg, anchors)
        assert reg.dtype == torch.float32 and reg.is_contiguous() and anchors.is_contiguous() and anchors.shape == (n, 7)
    import ctypes
    mv = ctypes.byref(ctypes.c_float(float(min_value))) if min_value is not None else None
    im = _lib.ints(tuple(int(v) for v in idx_map)) if idx_map is not None else None
    # key scratch: the selection evaluates every element once and re-reads the cached keys with wide loads
    nbytes = int(lib().d3d_topk_scratch_bytes(int(n), S)) if n >= 8192 else 0
    buf = None
    if nbytes:
        buf = _NMS_SCRATCH.get(("topk",) + _nms_key(dev))
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
            _NMS_SCRATCH[("topk",) + _nms_key(dev)] = buf
    check(lib().d3d_topk_segments(ptr(vals), int(n), int(elem_stride), int(group_stride), int(n_groups), ptr(example),
                                  int(n_examples), k, int(bool(sigmoid)), mv, im, ptr(reg),
                                  int(reg.shape[1]) if reg is not None else 0, ptr(anchors), float(clip), ptr(out["idx32"]),
                                  ptr(out["idx"]), ptr(out["scores"]), ptr(out["props"]), ptr(out["counts"]), ptr(buf),
                                  buf.numel() if buf is not None else 0, stream_of()))
    return out


def rotate_nms_3d(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5, flag='', aug_thickness=(0.0, 0.0)):
    """second/pytorch/core/box_torch_ops.py:489-514 through ONE library entry (d3d_rotate_nms_3d): rbboxes [n,7] yx_zb,
    scores [n] -> LongTensor of kept indices into the input, in score order (equal scores: lower index first).
    aug_thickness: the size clamp boxlist_nms_3d applies for the IoU only."""
    rbboxes, scores = _f32c(rbboxes), _f32c(scores)
    require_gpu(rbboxes, scores)
    n = rbboxes.shape[0]
    dev = rbboxes.device
    if n == 0:
        return torch.zeros([0], dtype=torch.int64, device=dev)
    k = n if pre_max_size is None else min(n, int(pre_max_size))
    if k > topk_max():
        raise D3DError(f"rotate_nms_3d: more than {topk_max()} candidates; pass pre_max_size (reference uses 2000)")
    nbytes = lib().d3d_rotate_nms_3d_scratch_bytes(k, n)
    buf = _NMS_SCRATCH.get(_nms_key(dev))
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _NMS_SCRATCH[_nms_key(dev)] = buf
    keep = torch.empty(k, dtype=torch.int64, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev)
    import ctypes
    n_host = ctypes.c_int(0)
    check(lib().d3d_rotate_nms_3d(ptr(rbboxes), ptr(scores), n, k, int(post_max_size or 0), float(iou_threshold),
                                  float(aug_thickness[0]), float(aug_thickness[1]), ptr(keep), ptr(nk),
                                  ctypes.byref(n_host), ptr(buf), buf.numel(), stream_of()))
    return keep[:n_host.value]


def nms_3d_clamped(bbox3d, scores, nms_thresh, nms_aug_thickness=None, max_proposals=-1, flag=''):
    """Tensor form of boxlist_nms_3d: clamps dy,dx >= aug[0] and dz >= aug[1] for the IoU only, keeps
    pre_max 2000, returns kept indices."""
    if nms_aug_thickness is None:
        nms_aug_thickness = [0, 0]
    if flag == 'rpn_post':
        assert max_proposals > 100, max_proposals
    elif flag == 'roi_post':
        assert max_proposals == -1
    else:
        raise NotImplementedError(flag)
    if max_proposals < 0:
        max_proposals = 500
    return rotate_nms_3d(bbox3d, scores, pre_max_size=2000, post_max_size=max_proposals, iou_threshold=nms_thresh,
                         flag=flag, aug_thickness=nms_aug_thickness)


_UNIT_WEIGHTS = None


def box_decode_rows(box_encodings, anchors, rows, bbox_xform_clip=10000.0):
    """box_decode(box_encodings[rows], anchors[rows]) (unit weights, one class) without the two gathers:
    d3d_box_decode_rows.  rows int64 [n] on the device."""
    global _UNIT_WEIGHTS
    enc, anc = _f32c(box_encodings), _f32c(anchors)
    require_gpu(enc, anc, rows)
    assert enc.shape[1] == 7 and anc.shape[1] == 7 and enc.shape[0] == anc.shape[0] and rows.dtype == torch.int64
    if _UNIT_WEIGHTS is None:
        _UNIT_WEIGHTS = floats((1.0,) * 7)
    out = torch.empty((rows.shape[0], 7), dtype=torch.float32, device=enc.device)
    check(lib().d3d_box_decode_rows(ptr(enc), ptr(anc), ptr(rows), rows.shape[0], _UNIT_WEIGHTS, float(bbox_xform_clip),
                                    ptr(out), stream_of()))
    return out


def gather_kept(boxes, scores, keep, n_keep, rows_out, min_size, count_host=None):
    """d3d_gather_kept: -> (boxes [rows_out, 7] with clamped sizes, scores [rows_out]) of the NMS survivors keep[:n_keep]
    (n_keep int32 [1] still on the device), padded with candidate 0.  count_host: pinned int32 [1] host tensor that
    receives n_keep by a store of the kernel (read it after an event recorded behind this call)."""
    if count_host is not None:
        assert count_host.dtype == torch.int32 and count_host.is_pinned()
    require_gpu(boxes, scores, keep, n_keep)
    assert boxes.dtype == torch.float32 and scores.dtype == torch.float32 and keep.dtype == torch.int32
    assert n_keep.dtype == torch.int32 and boxes.shape[0] > 0 and keep.numel() >= rows_out
    ob = torch.empty((rows_out, 7), dtype=torch.float32, device=boxes.device)
    os_ = torch.empty((rows_out,), dtype=torch.float32, device=boxes.device)
    check(lib().d3d_gather_kept(ptr(boxes), ptr(scores), ptr(keep), ptr(n_keep), rows_out, float(min_size), ptr(ob),
                                ptr(os_), ptr(count_host), stream_of()))
    return ob, os_


def box_decode(box_encodings, anchors, weights=(1.0,) * 7, bbox_xform_clip=10000.0):
    """BoxCoder3D.decode with smooth_dim=True; box_encodings [n, 7*nc], anchors [n, 7]."""
    enc, anc = _f32c(box_encodings), _f32c(anchors)
    require_gpu(enc, anc)
    assert enc.shape[0] == anc.shape[0] and anc.shape[1] == 7
    nc = enc.shape[1] // 7
    out = torch.empty_like(enc)
    if enc.shape[0] and nc != 1:        # every class of a row against the row's anchor, no repeated anchor tensor
        check(lib().d3d_box_decode_classes(ptr(enc), ptr(anc), enc.shape[0], nc, floats(weights), float(bbox_xform_clip),
                                           ptr(out), stream_of()))
    elif enc.shape[0]:
        check(lib().d3d_box_decode(ptr(enc), ptr(anc), enc.shape[0], floats(weights), float(bbox_xform_clip),
                                   ptr(out), stream_of()))
    return out

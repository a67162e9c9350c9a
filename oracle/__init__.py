"""TEST INFRASTRUCTURE ONLY: ctypes + numpy front end of the CPU oracle.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package `detection_3d_amd` never imports this module.

See scn_oracle.cpp / box_oracle.cpp for what is pinned and what is "parity unpinned".
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("scn_oracle.cpp", "box_oracle.cpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(
            ["g++", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c++17", "-ffp-contract=off"]
            + srcs + ["-o", so])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_subm_nbr.restype = ctypes.c_long
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------------------------------------
# a1: data3d/suncg_utils/suncg_dataset.py:97-177 (augmentations all off, :78-83)
def voxelize(pcl, scale=50, full_scale=(4096, 4096, 512)):
    """pcl float32 [N,9] -> (coords int64 [M,3], feats float32 [M,9]).  numpy, as the reference."""
    a = pcl[:, 0:3].copy()
    b = pcl.copy()
    m = np.eye(3) * scale            # float64, suncg_dataset.py:115-119
    a = np.matmul(a, m)              # float32 @ float64 -> float64, :123
    offset = -a.min(0)               # :127-133
    a += offset
    b[:, 0:3] = a / scale            # :149
    full = np.asarray(full_scale)
    up_check = np.all(a < full[np.newaxis, :], 1)   # :160
    idxs = (a.min(1) >= 0) * up_check               # :169
    a = a[idxs]
    b = b[idxs]
    return a.astype(np.int64), b     # torch .long() truncates like astype(int64), :173


def input_sites(coords):
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    n, ncols = coords.shape
    sop = np.empty(n, np.int32)
    loc = np.empty((max(n, 1), 4), np.int32)
    na = lib().orc_input_sites(_p(coords), n, ncols, _p(sop), _p(loc))
    return sop, loc[:na].copy()


def input_rule_table(site_of_point, n_active):
    sop = _i32(site_of_point)
    ma = lib().orc_input_rule_table(_p(sop), len(sop), n_active, None)
    rules = np.zeros((n_active, 1 + ma), np.int32)
    lib().orc_input_rule_table(_p(sop), len(sop), n_active, _p(rules))
    return rules


def input_forward(feats, site_of_point, n_active, average=True):
    feats = _f32(feats)
    sop = _i32(site_of_point)
    out = np.empty((n_active, feats.shape[1]), np.float32)
    lib().orc_input_forward(_p(feats), feats.shape[0], feats.shape[1], _p(sop), n_active,
                            int(average), _p(out))
    return out


def subm_nbr(loc, filt):
    loc = _i32(loc)
    filt = np.asarray(filt, np.int32)
    K = int(np.prod(filt))
    nbr = np.empty((loc.shape[0], K), np.int32)
    total = lib().orc_subm_nbr(_p(loc), loc.shape[0], _p(filt), _p(nbr))
    return nbr, int(total)


def conv_rules(loc, filt, stride, out_size):
    """Returns (loc_out [nOut,4], rules [R,3] = (in, out, offset))."""
    loc = _i32(loc)
    filt = np.asarray(filt, np.int32)
    stride = np.asarray(stride, np.int32)
    out_size = np.asarray(out_size, np.int32)
    max_out = int(np.prod(-(-filt // stride)))
    n = loc.shape[0]
    loc_out = np.empty((max(n * max_out, 1), 4), np.int32)
    rules = np.empty((max(n * max_out, 1), 3), np.int32)
    nr = ctypes.c_long(0)
    n_out = lib().orc_conv_rules(_p(loc), n, _p(filt), _p(stride), _p(out_size), _p(loc_out),
                                 _p(rules), ctypes.byref(nr))
    return loc_out[:n_out].copy(), rules[:nr.value].copy()


def rule_conv(feats, W, rules, n_out, deconv=False):
    """W [K, Cin, Cout]; rules (in,out,offset); deconv swaps roles (CPU/Deconvolution.cpp:29-37)."""
    feats = _f32(feats)
    W = _f32(W)
    K, cin, cout = W.shape
    rules = _i32(rules)
    if deconv:
        rules = np.ascontiguousarray(rules[:, [1, 0, 2]])
    out = np.empty((n_out, cout), np.float32)
    lib().orc_rule_conv(_p(feats), cin, _p(W), K, cout, _p(rules), ctypes.c_long(rules.shape[0]),
                        _p(out), n_out)
    return out


def nbr_conv(feats, W, nbr):
    feats = _f32(feats)
    W = _f32(W)
    K, cin, cout = W.shape
    nbr = _i32(nbr)
    out = np.empty((nbr.shape[0], cout), np.float32)
    lib().orc_nbr_conv(_p(feats), cin, _p(W), K, cout, _p(nbr), _p(out), nbr.shape[0])
    return out


def bn_forward(x, running_mean, running_var, weight, bias, eps, momentum, train, leakiness):
    x = _f32(x)
    n, c = x.shape
    out = np.empty_like(x)
    sm = np.zeros(c, np.float32)
    si = np.zeros(c, np.float32)
    rm = _f32(running_mean).copy()
    rv = _f32(running_var).copy()
    w = _f32(weight) if weight is not None else None
    b = _f32(bias) if bias is not None else None
    lib().orc_bn_forward(_p(x), _p(out), c, n, _p(sm), _p(si), _p(rm), _p(rv),
                         _p(w) if w is not None else None, _p(b) if b is not None else None,
                         ctypes.c_float(eps), ctypes.c_float(momentum), int(train),
                         ctypes.c_float(leakiness))
    return out, sm, si, rm, rv


def rule_conv_backward(feats, W, rules, d_out, deconv=False):
    """Returns (d_feats, dW[K,Cin,Cout]) of rule_conv (CPU/Convolution.cpp:81-115)."""
    feats, W, d_out = _f32(feats), _f32(W), _f32(d_out)
    K, cin, cout = W.shape
    rules = _i32(rules)
    if deconv:
        rules = np.ascontiguousarray(rules[:, [1, 0, 2]])
    d_in = np.empty_like(feats)
    dW = np.empty_like(W)
    lib().orc_rule_conv_backward(_p(feats), cin, _p(W), K, cout, _p(rules), ctypes.c_long(rules.shape[0]),
                                 _p(d_out), feats.shape[0], _p(d_in), _p(dW))
    return d_in, dW


def bn_backward(x, out, d_out, save_mean, save_invstd, weight, leakiness):
    """Returns (d_in, d_weight, d_bias) (CPU/BatchNormalization.cpp:62-107)."""
    x, out = _f32(x), _f32(out)
    d_out = _f32(d_out).copy()
    n, c = x.shape
    d_in = np.empty_like(x)
    dw = np.zeros(c, np.float32)
    db = np.zeros(c, np.float32)
    w = _f32(weight) if weight is not None else None
    lib().orc_bn_backward(_p(x), _p(d_in), _p(out), _p(d_out), c, n, _p(_f32(save_mean)), _p(_f32(save_invstd)),
                          _p(w) if w is not None else None, _p(dw), _p(db), ctypes.c_float(leakiness))
    return d_in, dw, db


def input_backward(d_out, site_of_point, n_active, average=True):
    d_out = _f32(d_out)
    sop = _i32(site_of_point)
    d_in = np.empty((len(sop), d_out.shape[1]), np.float32)
    lib().orc_input_backward(_p(d_out), d_out.shape[1], _p(sop), len(sop), n_active, int(average), _p(d_in))
    return d_in


def roi_align_rotated_3d_backward(top_diff, rois, spatial_scale, ph, pw, pz, sampling_ratio, dense_shape):
    top_diff, rois = _f32(top_diff), _f32(rois)
    B, C, H, W, Z = dense_shape
    out = np.zeros(dense_shape, np.float32)
    lib().orc_roi_align_rotated_3d_backward(_p(top_diff), B, C, H, W, Z, _p(rois), rois.shape[0],
                                            ctypes.c_float(spatial_scale), ph, pw, pz, sampling_ratio, _p(out))
    return out


def sparse_to_dense(feats, loc, size, batch=1):
    feats = _f32(feats)
    loc = _i32(loc)
    size = np.asarray(size, np.int32)
    out = np.empty((batch, feats.shape[1], size[0], size[1], size[2]), np.float32)
    lib().orc_sparse_to_dense(_p(feats), feats.shape[1], _p(loc), loc.shape[0], _p(size), batch,
                              _p(out))
    return out


# ---------------------------------------------------------------------------------------------
def rotate_iou_eval(boxes, query, criterion=-1):
    boxes = _f32(boxes)
    query = _f32(query)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    if out.size:
        lib().orc_rotate_iou_eval(_p(boxes), boxes.shape[0], _p(query), query.shape[0],
                                  int(criterion), _p(out))
    return out


def rotate_iou_raw(boxes, query, criterion=-1, eps_variant=False):
    """rotate_iou_gpu_eval without check_same_boxes; eps_variant=True additionally uses the
    commented-out tolerance predicate of nms_gpu.py:326-327 (pins test_nms_gpu.py:14-15)."""
    boxes = _f32(boxes)
    query = _f32(query)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    lib().orc_rotate_iou_raw(_p(boxes), boxes.shape[0], _p(query), query.shape[0], int(criterion),
                             int(eps_variant), _p(out))
    return out


def boxes_iou_3d(targets, anchors, aug=None, criterion=-1, only_xy=False):
    """aug = dict(target_Y, target_Z, anchor_Y, anchor_Z) or None."""
    targets = _f32(targets)
    anchors = _f32(anchors)
    if aug is None:
        aug = dict(target_Y=0.0, target_Z=0.0, anchor_Y=0.0, anchor_Z=0.0)
    a = np.array([aug["target_Y"], aug["target_Z"], aug["anchor_Y"], aug["anchor_Z"]], np.float32)
    out = np.zeros((targets.shape[0], anchors.shape[0]), np.float32)
    if out.size:
        lib().orc_boxes_iou_3d(_p(targets), targets.shape[0], _p(anchors), anchors.shape[0], _p(a),
                               int(criterion), int(only_xy), _p(out))
    return out


def rotate_nms_3d(boxes, scores, thresh):
    """Greedy NMS on already clamped / top-k'd boxes; returns kept indices (selection order)."""
    boxes = _f32(boxes)
    scores = _f32(scores)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), np.int32)
    nk = lib().orc_rotate_nms_3d(_p(boxes), _p(scores), n, ctypes.c_float(thresh), _p(keep))
    return keep[:nk].astype(np.int64)


def box_decode(enc, anchors, weights=(1.0,) * 7, clip=10000.0):
    enc = _f32(enc)
    anchors = _f32(anchors)
    w = np.asarray(weights, np.float32)
    out = np.empty_like(enc)
    lib().orc_box_decode(_p(enc), _p(anchors), enc.shape[0], _p(w), ctypes.c_float(clip), _p(out))
    return out


def roi_align_rotated_3d(inp, rois, spatial_scale, ph, pw, pz, sampling_ratio):
    inp = _f32(inp)
    rois = _f32(rois)
    B, C, H, W, Z = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw, pz), np.float32)
    if K:
        lib().orc_roi_align_rotated_3d(_p(inp), B, C, H, W, Z, _p(rois), K,
                                       ctypes.c_float(spatial_scale), ph, pw, pz, sampling_ratio,
                                       _p(out))
    return out

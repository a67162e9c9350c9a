"""ctypes binding of libd3d_hip.so (include/d3d_hip.h).  There is NO CPU fallback: if the HIP
library is missing or fails to load, importing an op raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D3D_LIB_PATH") or os.path.join(_HERE, "lib", "libd3d_hip.so")   # (the override: A/B runs of library builds)

_lib = None

c_int_p = ctypes.POINTER(ctypes.c_int)
c_float_p = ctypes.POINTER(ctypes.c_float)
vp = ctypes.c_void_p


class BnPrologue(ctypes.Structure):
    """d3d_bn_prologue (include/d3d_hip.h): device pointers of the producer's BatchNorm."""
    _fields_ = [("mean", vp), ("invstd", vp), ("weight", vp), ("bias", vp), ("leakiness", ctypes.c_float),
                ("out_stats", vp), ("out_stats_cap", ctypes.c_int), ("out_stats_rows", ctypes.POINTER(ctypes.c_int))]


bn_p = ctypes.POINTER(BnPrologue)

_SIGS = {
    "d3d_last_error": (ctypes.c_char_p, []),
    "d3d_abi_version": (ctypes.c_int, []),
    "d3d_meta_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_size_t]),
    "d3d_meta_destroy": (ctypes.c_int, [vp]),
    "d3d_meta_clear": (ctypes.c_int, [vp]),
    "d3d_meta_arena_used": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_size_t)]),
    "d3d_meta_set_geometry_stream": (ctypes.c_int, [vp, vp, ctypes.c_int]),
    "d3d_meta_set_plan_stream": (ctypes.c_int, [vp, vp, ctypes.c_int]),
    "d3d_geometry_async_start": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp]),
    "d3d_geometry_async_wait": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), vp]),
    "d3d_geometry_async_finish": (ctypes.c_int, [vp]),
    "d3d_grid_chain_enable": (ctypes.c_int, [ctypes.c_int]),
    "d3d_conv_ws_mode": (ctypes.c_int, [ctypes.c_int]),
    "d3d_conv_late_mode": (ctypes.c_int, [ctypes.c_int]),
    "d3d_conv_dw_deterministic": (ctypes.c_int, [ctypes.c_int]),
    "d3d_grid_chain_head": (ctypes.c_int, [ctypes.c_int]),
    "d3d_sort_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "d3d_sort_pairs": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t, vp]),
    "d3d_conv_time_next": (ctypes.c_int, [vp, vp]),
    "d3d_input_layer_prepare": (ctypes.c_int, [vp, vp]),
    "d3d_post_scores": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp]),
    "d3d_post_order": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_post_gather": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    "d3d_post_select_max": (ctypes.c_int, []),
    "d3d_post_select": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp,
                                       vp]),
    "d3d_roi_prepare": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_float, c_float_p, ctypes.c_int, ctypes.c_float, vp, vp,
                                       vp, vp]),
    "d3d_roi_prepare_counted": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_float, c_float_p, ctypes.c_int,
                                               ctypes.c_float, vp, vp, vp, vp]),
    "d3d_voxelize": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_int_p, vp, vp,
                                    c_int_p, vp, ctypes.c_size_t, vp]),
    "d3d_voxelize_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "d3d_input_layer_build": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_int,
                                             ctypes.c_int, vp, c_int_p]),
    "d3d_input_layer_build_prefetch": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_int,
                                                      ctypes.c_int, c_int_p, vp, c_int_p]),
    "d3d_input_layer_forward": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp]),
    "d3d_input_layer_export": (ctypes.c_int, [vp, vp, vp, vp]),
    "d3d_get_n_active": (ctypes.c_int, [vp, c_int_p, c_int_p]),
    "d3d_get_spatial_locations": (ctypes.c_int, [vp, c_int_p, vp, vp]),
    "d3d_subm_prepare": (ctypes.c_int, [vp, c_int_p, c_int_p, vp, ctypes.POINTER(ctypes.c_long)]),
    "d3d_conv_prepare": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, c_int_p,
                                        ctypes.POINTER(ctypes.c_long)]),
    "d3d_deconv_prepare": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp,
                                          ctypes.POINTER(ctypes.c_long)]),
    "d3d_export_rules": (ctypes.c_int, [vp, ctypes.c_int, c_int_p, c_int_p, c_int_p, vp, ctypes.c_long,
                                        ctypes.POINTER(ctypes.c_long), vp]),
    "d3d_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "d3d_pack_conv_weight": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_subm_conv_forward": (ctypes.c_int, [vp, c_int_p, c_int_p, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                             vp, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_conv_forward": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp,
                                        ctypes.c_int, vp, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_deconv_forward": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp,
                                          ctypes.c_int, vp, vp, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_bn_apply": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_float, vp]),
    "d3d_bn_batch_invstd": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp,
                                           ctypes.c_size_t, vp]),
    "d3d_pack_conv_weight_transposed": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       vp, vp]),
    "d3d_subm_conv_backward": (ctypes.c_int, [vp, c_int_p, c_int_p, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp,
                                              vp, vp]),
    "d3d_conv_backward": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp,
                                         ctypes.c_int, vp, vp, vp, vp]),
    "d3d_deconv_backward": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp,
                                           ctypes.c_int, vp, vp, vp, vp]),
    "d3d_bn_backward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp,
                                       ctypes.c_float, vp, ctypes.c_size_t, vp]),
    "d3d_bn_backward_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "d3d_input_layer_backward": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp]),
    "d3d_sparse_to_dense_backward": (ctypes.c_int, [vp, c_int_p, vp, ctypes.c_int, vp, vp]),
    "d3d_roi_align_rotated_3d_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_int, vp, ctypes.c_int, ctypes.c_float,
                                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         vp, vp]),
    "d3d_roi_align_rotated_3d_sparse_backward": (ctypes.c_int, [vp, c_int_p, vp, ctypes.c_int, c_int_p, vp,
                                                                ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                                                ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                vp, vp]),
    "d3d_bn_forward": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float, vp,
                                      ctypes.c_size_t, vp]),
    "d3d_bn_batch_stats": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t, vp]),
    "d3d_bn_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "d3d_add": (ctypes.c_int, [vp, vp, vp, ctypes.c_size_t, vp]),
    "d3d_sparse_to_dense_forward": (ctypes.c_int, [vp, c_int_p, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_roi_align_rotated_3d_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, vp, ctypes.c_int,
                                                        ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_roi_align_rotated_3d_sparse_forward": (ctypes.c_int, [vp, c_int_p, vp, ctypes.c_int, c_int_p, vp,
                                                               ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                                               ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                               vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_roi_align_rotated_3d_sparse_forward_levels": (ctypes.c_int, [vp, ctypes.c_int, c_int_p, ctypes.POINTER(vp),
                                                                      ctypes.c_int, c_float_p, vp, ctypes.c_int,
                                                                      ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                      ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "d3d_rotate_iou_eval": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_boxes_iou_3d": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                        ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_rotate_nms_3d_batched": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp, vp,
                                                 ctypes.c_size_t, vp]),
    "d3d_nms_batched_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "d3d_rotate_nms_3d": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, vp, vp, c_int_p, vp, ctypes.c_size_t, vp]),
    "d3d_rotate_nms_3d_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "d3d_topk_segments": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, c_float_p, c_int_p, vp, ctypes.c_int, vp,
                                         ctypes.c_float, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "d3d_topk_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "d3d_topk_max": (ctypes.c_int, []),
    "d3d_anchors": (ctypes.c_int, [vp, c_int_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_float), ctypes.c_float, vp, vp]),
    "d3d_anchors_maps": (ctypes.c_int, [vp, ctypes.c_int, c_int_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_float), ctypes.c_float, vp, vp]),
    "d3d_rpn_head": (ctypes.c_int, [ctypes.POINTER(vp), c_int_p, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp,
                                    ctypes.c_int, vp, vp, vp]),
    "d3d_mlp_heads": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp]),
    "d3d_rotate_nms_3d_sorted": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, vp, vp, vp,
                                                ctypes.c_size_t, vp]),
    "d3d_nms_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "d3d_box_decode": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_float,
                                      vp, vp]),
    "d3d_box_decode_classes": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                              ctypes.c_float, vp, vp]),
    "d3d_box_decode_rows": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_float,
                                           vp, vp]),
    "d3d_gather_kept": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_float, vp, vp, vp, vp]),
    "d3d_plan_stats": (ctypes.c_int, [vp, ctypes.c_int, c_int_p, c_int_p, c_int_p, ctypes.POINTER(ctypes.c_long),
                                      ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), vp]),
    # storage-type aware forms (d3d_dtype: 0 fp32, 1 bf16)
    "d3d_packed_weight_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "d3d_pack_conv_weight_dt": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]),
    "d3d_subm_conv_forward_dt": (ctypes.c_int, [vp, c_int_p, c_int_p, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp,
                                                ctypes.c_int, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_conv_forward_dt": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp, ctypes.c_int,
                                           vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_deconv_forward_dt": (ctypes.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, vp, ctypes.c_int, vp, ctypes.c_int,
                                             vp, vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_double), bn_p]),
    "d3d_conv_bf16_tuning": (ctypes.c_int, [ctypes.c_int, ctypes.c_long]),
    "d3d_bn_stats_from_partials": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                                  vp, vp, vp, ctypes.c_size_t, vp]),
    "d3d_bn_batch_invstd_dt": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp,
                                              ctypes.c_size_t, ctypes.c_int, vp]),
    "d3d_rows_to_bf16": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, vp]),
    "d3d_bn_apply_dt": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_float,
                                       ctypes.c_int, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


class D3DError(RuntimeError):
    pass


def lib():
    """Loads the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise D3DError(
                f"{LIB_PATH} not found: build it with `python -m detection_3d_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().d3d_last_error()
        raise D3DError(f"libd3d_hip error {rc}: {msg.decode() if msg else ''}")


_INTS = {}     # tuple -> ctypes int array; the library only reads these arrays


def ints(values):
    if type(values) is tuple:
        a = _INTS.get(values)
        if a is None:
            if len(_INTS) > 4096:
                _INTS.clear()
            a = _INTS[values] = (ctypes.c_int * len(values))(*values)
        return a
    values = [int(v) for v in values]
    return (ctypes.c_int * len(values))(*values)


def floats(values):
    values = [float(v) for v in values]
    return (ctypes.c_float * len(values))(*values)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream(device=None):
    """hipStream_t of torch's current stream as an int (torch.cuda.current_stream costs ~10 us of Python per
    call, and every library call needs the stream)."""
    if _RAW_STREAM is None or _GET_DEVICE is None:
        return torch.cuda.current_stream(device).cuda_stream
    idx = None if device is None else (device if isinstance(device, int) else device.index)
    return _RAW_STREAM(_GET_DEVICE() if idx is None else idx)


def stream_of(device=None):
    return ctypes.c_void_p(raw_stream(device))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise D3DError("this op runs on the MI355X only: tensor is on %s (no CPU fallback)" % t.device)
        if t is not None and not t.is_contiguous():
            raise D3DError("tensor must be contiguous")


_HOST_WORDS = {}
_HOST_WORD_RING = 8
_HOST_WORD_BLOCKING = os.environ.get("D3D_COUNT_EVENT_BLOCKING", "1") != "0"


def host_word(device, tag):
    """-> (pinned int32 [1] host tensor, event): the next slot of a small ring kept per (device, current stream, tag), so
    that a deferred read-back (PaddedProposals.resolve) still finds its own word and event when further counts have been
    requested on the same stream meanwhile (up to _HOST_WORD_RING outstanding ones).  A kernel stores a count to the word
    (pinned host memory is device-visible at the same address), the caller records the event behind that launch and
    the host waits for the event alone: no copy engine between the kernel and the host, and launches enqueued behind the
    event do not hold the host up."""
    key = (device.index, raw_stream(device), tag)
    ring = _HOST_WORDS.get(key)
    if ring is None:
        ring = _HOST_WORDS[key] = [[], 0]
    slots, nxt = ring
    if len(slots) < _HOST_WORD_RING:
        slots.append((torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(blocking=_HOST_WORD_BLOCKING)))
        return slots[-1]
    ring[1] = (nxt + 1) % _HOST_WORD_RING
    return slots[nxt]

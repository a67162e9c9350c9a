"""The C-ABI library loads (no GPU needed) and exports every symbol include/d3d_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "d3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from detection_3d_amd import _lib
    from detection_3d_amd.build import build_library
    build_library()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(handle, n), f"{n} declared in d3d_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), set(names) ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.lib()
    assert lib.d3d_abi_version() == 1
    # pure host-side helpers are callable without a GPU
    assert lib.d3d_packed_weight_floats(27, 9, 32) == 27 * 16 * 32
    assert lib.d3d_packed_weight_floats(8, 128, 128) == 8 * 128 * 128
    assert lib.d3d_packed_weight_floats(1, 300, 128) == 0
    assert lib.d3d_nms_scratch_bytes(2000) >= 2000 * 32 * 8


def test_ops_refuse_cpu_tensors():
    import torch
    from detection_3d_amd import _lib, box_ops
    with pytest.raises(_lib.D3DError):
        box_ops.rotate_iou_gpu_eval(torch.zeros(2, 5), torch.zeros(2, 5))

"""Host-side helpers of the operator mirror (CPU): cached size tuples / ctypes arrays, the no-grad bypass of
autograd.Function.apply, the geometry schedule of FPN_Net (which level builds which grid / rulebook)."""
import pytest
import torch

from detection_3d_amd import _lib
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_backbone
from detection_3d_amd.sparseconvnet import SCN, modules


def test_size3_and_ints_are_cached_and_exact():
    t = torch.LongTensor([4096, 4096, 512])
    a = SCN._size3(t)
    assert a == (4096, 4096, 512) and SCN._size3(t) is a            # remembered on the tensor
    assert SCN._size3([2, 2, 2]) == (2, 2, 2) and SCN._size3((1, 1, 4)) == (1, 1, 4)
    out = (t - torch.LongTensor([2, 2, 2])) // torch.LongTensor([2, 2, 2]) + 1
    assert SCN._size3(out) == (2048, 2048, 256)                      # a new tensor gets its own tuple
    x, y = _lib.ints((3, 3, 3)), _lib.ints((3, 3, 3))
    assert x is y and list(x) == [3, 3, 3] and list(_lib.ints([0, 1, 2])) == [0, 1, 2]


def test_apply_bypasses_autograd_only_without_grad():
    class Twice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    x = torch.ones(3, requires_grad=True)
    y = modules._apply(Twice, x)
    assert y.requires_grad and y.grad_fn is not None
    y.sum().backward()
    assert torch.equal(x.grad, torch.full((3,), 2.0))
    with torch.no_grad():
        z = modules._apply(Twice, x)
    assert not z.requires_grad and torch.equal(z, torch.full((3,), 2.0))


def test_geometry_schedule_of_the_4c_backbone(monkeypatch):
    """_geometry_steps: level k builds the strided grid k-1 -> k, the z-collapsing projection of the RPN maps that live
    at level k and (full) the 3x3x3 / lateral 1x1x1 / deconvolution rulebooks the level's convolutions use."""
    cfg = get_cfg("4c_Fpn432")
    net = build_backbone(cfg)
    calls = []
    monkeypatch.setattr(SCN, "Convolution_prepare", lambda i, o, f, s, m: calls.append(("grid", SCN._size3(i), SCN._size3(o), SCN._size3(f))) or 1)
    monkeypatch.setattr(SCN, "SubmanifoldConvolution_prepare", lambda sz, f, m: calls.append(("subm", SCN._size3(sz), SCN._size3(f))))
    monkeypatch.setattr(SCN, "Deconvolution_prepare", lambda i, o, f, s, m: calls.append(("deconv", SCN._size3(i), SCN._size3(o))))

    class Net0(object):
        spatial_size = torch.LongTensor(cfg.SPARSE3D.VOXEL_FULL_SCALE)
        metadata = None

    per_level = []
    for k in net._geometry_steps(Net0(), True):
        per_level.append(list(calls))
        calls.clear()
    n_scales = len(net.m_downs)
    assert len(per_level) == n_scales
    sizes = [tuple(cfg.SPARSE3D.VOXEL_FULL_SCALE)]
    for k in range(1, n_scales):
        sizes.append(tuple(s // 2 for s in sizes[-1]))
    rpn_levels = {n_scales - 1 - s for s in cfg.MODEL.RPN.RPN_SCALES_FROM_TOP}
    lowest_up = n_scales - 1 - max(list(cfg.MODEL.RPN.RPN_SCALES_FROM_TOP) + list(cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_FROM_TOP))
    for k, got in enumerate(per_level):
        grids = [c for c in got if c[0] == "grid"]
        if k > 0:
            assert ("grid", sizes[k - 1], sizes[k], (2, 2, 2)) in grids
        proj = [c for c in grids if c[3][:2] == (1, 1)]
        assert len(proj) == (1 if k in rpn_levels else 0)
        if proj:
            assert proj[0][1] == sizes[k] and proj[0][2] == (sizes[k][0], sizes[k][1], 1)
        assert ("subm", sizes[k], (3, 3, 3)) in got
        assert (("subm", sizes[k], (1, 1, 1)) in got) == (k >= lowest_up)
        assert (("deconv", sizes[k], sizes[k - 1]) in got) == (k > lowest_up)
    # the plain pre-pass builds grids only
    calls.clear()
    net.prepare_geometry(Net0(), full=False)
    assert calls and all(c[0] == "grid" for c in calls)


@pytest.mark.parametrize("name", ["4c_Fpn432", "3G6c_Fpn4321"])
def test_async_geometry_specs_are_the_grid_calls_of_the_schedule(monkeypatch, name):
    """FPN_Net._geometry_specs -- the list the library's geometry thread works through (d3d_geometry_async_start) -- holds
    exactly the d3d_conv_prepare calls _geometry_steps makes, in the same order, and `last[k]` points at the last one a
    level needs before its convolutions may be enqueued."""
    cfg = get_cfg(name)
    net = build_backbone(cfg)
    calls = []
    monkeypatch.setattr(SCN, "Convolution_prepare",
                        lambda i, o, f, s, m: calls.append(SCN._size3(i) + SCN._size3(o) + SCN._size3(f) + SCN._size3(s)) or 1)

    class Net0(object):
        spatial_size = torch.LongTensor(cfg.SPARSE3D.VOXEL_FULL_SCALE)
        metadata = None

    per_level = []
    for k in net._geometry_steps(Net0(), False):
        per_level.append(len(calls))
    specs, last, last_all = net._geometry_specs(Net0.spatial_size)
    assert all(s[0] == 1 for s in specs) and [tuple(s[1:]) for s in specs] == calls
    assert len(last) == len(net.m_downs) and last_all == len(specs) - 1
    for k, n_after in enumerate(per_level):
        before = per_level[k - 1] if k else 0
        assert last[k] == ((n_after - 1 if n_after > before else -1), -1)
    # with the views: the same grids in the same order, every level waits for its 3x3x3 rulebook (listed right behind
    # its grids), and the rulebooks of the top-down path -- the `full` ones of _geometry_steps -- come last
    v, vlast, vall = net._geometry_specs(Net0.spatial_size, views=True)
    assert [tuple(s[1:]) for s in v if s[0] == 1] == calls and vall == len(v) - 1
    full = []
    monkeypatch.setattr(SCN, "SubmanifoldConvolution_prepare", lambda sz, f, m: full.append((0, SCN._size3(sz), SCN._size3(f))))
    monkeypatch.setattr(SCN, "Deconvolution_prepare", lambda i, o, f, s, m: full.append((2, SCN._size3(i), SCN._size3(o))))
    for k in net._geometry_steps(Net0(), True):
        pass
    got = [(s[0], tuple(s[1:4]), tuple(s[7:10])) if s[0] == 0 else (2, tuple(s[1:4]), tuple(s[4:7])) for s in v if s[0] != 1]
    size0 = tuple(cfg.SPARSE3D.VOXEL_FULL_SCALE)
    assert sorted(got + [(0, size0, (3, 3, 3))]) == sorted(full)      # (level 0's 3x3x3 is the caller's)
    for k, (grid_row, idx) in enumerate(vlast):
        if k == 0:
            assert idx == -1
            continue
        assert v[idx][0] == 0 and tuple(v[idx][7:10]) == (3, 3, 3) and v[grid_row][0] == 1 and grid_row < idx
        assert all(s[0] == 1 or tuple(s[7:10]) == (3, 3, 3) for s in v[:idx + 1])
        assert all(s[0] != 1 for s in v[grid_row + 1:idx + 1])          # the level's last grid row

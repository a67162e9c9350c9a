"""GPU parity of the backward ops (training rows) vs the CPU oracle restatements
(SCN/CPU/Convolution.cpp:81-115, BatchNormalization.cpp:62-107, IOLayers.cpp:30-47,
ROIAlignRotated3D_cuda.cu:182-354)."""
import numpy as np
import pytest
import torch

import oracle
from tests.helpers import small_scene

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _sparse_input(dev, cin, seed=4, n_points=6000, size=(64, 64, 16)):
    from detection_3d_amd import sparseconvnet as scn
    rng = np.random.RandomState(seed + cin)
    _, coords, _ = small_scene(seed, n_points, (1.2, 1.0, 0.3), size)
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    layer = scn.InputLayer(3, size, mode=4)
    ft = torch.from_numpy(feats).to(dev).requires_grad_(True)
    t = layer([torch.from_numpy(coords), ft])
    sop, loc = oracle.input_sites(coords)
    x = oracle.input_forward(feats, sop, loc.shape[0], True)
    return t, ft, x, sop, loc


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (256, 128), (128, 256), (256, 256)])
def test_conv_backward(dev, cin, cout):
    from detection_3d_amd import sparseconvnet as scn
    t, ft, x, sop, loc = _sparse_input(dev, cin)
    torch.manual_seed(1)
    rng = np.random.RandomState(7)
    # submanifold 3^3
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    y = conv(t).features
    g = rng.randn(*y.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(dev))
    w = conv.weight.detach().cpu().numpy().reshape(27, cin, cout)
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    from tests.helpers import nbr_to_rules
    rules = nbr_to_rules(nbr)
    d_x, d_w = oracle.rule_conv_backward(x, w, rules, g)
    assert rel(conv.weight.grad.cpu().numpy().reshape(27, cin, cout), d_w) < 2e-4
    d_feat_in = oracle.input_backward(d_x, sop, loc.shape[0], True)       # chain through the input layer
    assert rel(ft.grad.cpu().numpy(), d_feat_in) < 2e-4
    # strided conv 2/2 then deconvolution back
    ft.grad = None
    down = scn.Convolution(3, cin, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
    up = scn.Deconvolution(3, cout, cin, [2, 2, 2], [2, 2, 2], False).to(dev)
    d = down(t)
    u = up(d).features
    g2 = rng.randn(*u.shape).astype(np.float32)
    u.backward(torch.from_numpy(g2).to(dev))
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], [32, 32, 8])
    wd = down.weight.detach().cpu().numpy().reshape(8, cin, cout)
    wu = up.weight.detach().cpu().numpy().reshape(8, cout, cin)
    dmid = oracle.rule_conv(x, wd, ru, lo.shape[0])
    d_mid, d_wu = oracle.rule_conv_backward(dmid, wu, ru, g2, deconv=True)
    d_x2, d_wd = oracle.rule_conv_backward(x, wd, ru, d_mid)
    assert rel(up.weight.grad.cpu().numpy().reshape(8, cout, cin), d_wu) < 2e-4
    assert rel(down.weight.grad.cpu().numpy().reshape(8, cin, cout), d_wd) < 2e-4
    assert rel(ft.grad.cpu().numpy(), oracle.input_backward(d_x2, sop, loc.shape[0], True)) < 2e-4


def test_first_layer_weight_grad(dev):
    # Cin = 9: only dWeight exists (the input features need no gradient)
    from detection_3d_amd import sparseconvnet as scn
    from tests.helpers import nbr_to_rules
    t, ft, x, sop, loc = _sparse_input(dev, 9)
    t.features = t.features.detach()
    conv = scn.SubmanifoldConvolution(3, 9, 32, 3, False).to(dev)
    y = conv(t).features
    g = np.random.RandomState(2).randn(*y.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(dev))
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    _, d_w = oracle.rule_conv_backward(x, conv.weight.detach().cpu().numpy().reshape(27, 9, 32), nbr_to_rules(nbr), g)
    assert rel(conv.weight.grad.cpu().numpy().reshape(27, 9, 32), d_w) < 2e-4


def test_batchnorm_backward(dev):
    from detection_3d_amd import sparseconvnet as scn
    rng = np.random.RandomState(5)
    for C, leak in ((32, 0.0), (128, 0.333), (256, 0.0)):
        x = (rng.randn(4000, C) * 2 + 0.5).astype(np.float32)
        xt = torch.from_numpy(x).to(dev).requires_grad_(True)
        bn = scn.BatchNormLeakyReLU(C, momentum=0.95, leakiness=leak).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        y = bn(scn.SparseConvNetTensor(xt, None, torch.tensor([8, 8, 8]))).features
        g = rng.randn(*x.shape).astype(np.float32)
        y.backward(torch.from_numpy(g).to(dev))
        out, sm, si, _, _ = oracle.bn_forward(x, np.zeros(C), np.ones(C), bn.weight.detach().cpu().numpy(),
                                              bn.bias.detach().cpu().numpy(), 1e-4, 0.95, True, leak)
        d_in, d_w, d_b = oracle.bn_backward(x, out, g, sm, si, bn.weight.detach().cpu().numpy(), leak)
        assert rel(xt.grad.cpu().numpy(), d_in) < 5e-4
        assert rel(bn.weight.grad.cpu().numpy(), d_w) < 5e-4
        assert rel(bn.bias.grad.cpu().numpy(), d_b) < 5e-4


def test_roi_align_sparse_backward(dev):
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.roi_align_rotated_3d import roi_align_rotated_3d_sparse
    t, ft, x, sop, loc = _sparse_input(dev, 128, seed=8, n_points=9000)
    feats = t.features.detach().requires_grad_(True)
    t2 = scn.SparseConvNetTensor(feats, t.metadata, t.spatial_size)
    crop = (loc[:, :3].max(0) + 1).tolist()
    rng = np.random.RandomState(1)
    K = 40
    rois = np.zeros((K, 8), np.float32)
    rois[:, 1] = rng.rand(K) * crop[1] * 8
    rois[:, 2] = rng.rand(K) * crop[0] * 8
    rois[:, 3] = rng.rand(K) * crop[2] * 8
    rois[:, 4] = 4 + rng.rand(K) * 200
    rois[:, 5] = 2 + rng.rand(K) * 40
    rois[:, 6] = 4 + rng.rand(K) * 100
    rois[:, 7] = rng.rand(K) * 180
    rois[:4, 3] += 200                                        # z above the map: forward clamps, backward drops
    out = roi_align_rotated_3d_sparse(t2, torch.from_numpy(rois).to(dev), 1.0 / 8, 6, 8, 4, 2)
    g = rng.randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    dense = oracle.roi_align_rotated_3d_backward(g, rois, 1.0 / 8, 6, 8, 4, 2, (1, 128, crop[0], crop[1], crop[2]))
    want = dense[0][:, loc[:, 0], loc[:, 1], loc[:, 2]].T
    assert rel(feats.grad.cpu().numpy(), want) < 1e-4
    assert np.abs(want).max() > 0


def test_dense_roi_align_and_sparse_to_dense_backward(dev):
    """The reference's own composition: SparseToDense -> _C.roi_align_rotated_3d (dense), both differentiable;
    must equal the sparse-sampled op's gradient."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.roi_align_rotated_3d import roi_align_rotated_3d_forward, roi_align_rotated_3d_sparse
    t, ft, x, sop, loc = _sparse_input(dev, 64, seed=9, n_points=5000, size=(32, 32, 8))
    rng = np.random.RandomState(3)
    K = 25
    rois = np.zeros((K, 8), np.float32)
    rois[:, 1:4] = rng.rand(K, 3) * np.array([32, 32, 8]) * 4
    rois[:, 4:7] = 4 + rng.rand(K, 3) * np.array([60, 30, 20])
    rois[:, 7] = rng.rand(K) * 180
    r = torch.from_numpy(rois).to(dev)
    g = torch.from_numpy(rng.randn(K, 64, 6, 8, 4).astype(np.float32)).to(dev)
    grads = []
    for mode in ("dense", "sparse"):
        feats = t.features.detach().clone().requires_grad_(True)
        tt = scn.SparseConvNetTensor(feats, t.metadata, t.spatial_size)
        if mode == "dense":
            dense = scn.SparseToDense(3, 64)(tt, batch_size=1)
            out = roi_align_rotated_3d_forward(dense, r, 0.25, 6, 8, 4, 2)
        else:
            out = roi_align_rotated_3d_sparse(tt, r, 0.25, 6, 8, 4, 2, crop=[32, 32, 8])
        out.backward(g)
        grads.append((out.detach().cpu().numpy(), feats.grad.cpu().numpy()))
    assert rel(grads[0][0], grads[1][0]) < 1e-5
    assert rel(grads[0][1], grads[1][1]) < 1e-4
    want = oracle.roi_align_rotated_3d_backward(g.cpu().numpy(), rois, 0.25, 6, 8, 4, 2, (1, 64, 32, 32, 8))
    assert rel(grads[0][1], want[0][:, loc[:, 0], loc[:, 1], loc[:, 2]].T) < 1e-4


@pytest.mark.parametrize("cin,cout,n_points", [(32, 32, 60000), (64, 64, 60000), (128, 128, 20000), (64, 128, 3000), (256, 256, 300)])
def test_conv_backward_deterministic_dw(dev, cin, cout, n_points):
    """d3d_conv_dw_deterministic(1): dWeight without atomics -- the same bits in every run (the default accumulates with
    fp32 atomics like the reference, SCN/CUDA/Convolution.cu:249-442, and differs in the last bits from run to run), equal
    to the atomic form up to summation order (1e-5 of the tensor's magnitude) and to the oracle like it (2e-4).
    Submanifold 3x3x3, strided 2/2 and its deconvolution; plans of one run and of several runs per workgroup."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd._lib import lib
    from tests.helpers import nbr_to_rules
    size = (128, 128, 32)
    rng = np.random.RandomState(cin + cout)
    _, coords, _ = small_scene(5, n_points, (2.5, 2.0, 0.6), size)
    feats = torch.from_numpy(rng.randn(coords.shape[0], cin).astype(np.float32)).to(dev)
    torch.manual_seed(3)
    sub = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    down = scn.Convolution(3, cin, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
    up = scn.Deconvolution(3, cout, cin, [2, 2, 2], [2, 2, 2], False).to(dev)

    def grads():
        for m in (sub, down, up):
            m.weight.grad = None
        t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), feats])
        y = sub(t).features
        y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
        u = up(down(t)).features
        u.backward(torch.ones_like(u) * 0.25 + u.detach() * 0.1)
        return [m.weight.grad.clone() for m in (sub, down, up)], t

    atomic, t = grads()
    was = lib().d3d_conv_dw_deterministic(1)
    try:
        a, _ = grads()
        b, _ = grads()
    finally:
        lib().d3d_conv_dw_deterministic(was)
    for x, y, z in zip(a, b, atomic):
        assert torch.isfinite(x).all() and float(x.abs().max()) > 0
        assert torch.equal(x, y)                                              # bit-stable
        assert float((x - z).abs().max()) <= 1e-5 * float(z.abs().max())      # the atomic form, another order
    # ... and the oracle, for the submanifold layer
    sop, loc = oracle.input_sites(coords)
    xin = oracle.input_forward(feats.cpu().numpy(), sop, loc.shape[0], True)
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    w = sub.weight.detach().cpu().numpy().reshape(27, cin, cout)
    rules = nbr_to_rules(nbr)
    yo = oracle.rule_conv(xin, w, rules, loc.shape[0])
    g = (0.5 + 0.1 * yo).astype(np.float32)
    _, d_w = oracle.rule_conv_backward(xin, w, rules, g)
    assert rel(a[0].cpu().numpy().reshape(27, cin, cout), d_w) < 2e-4

"""bf16 storage of the sparse-conv path (BASELINE.json configs[4]: "bf16 sparse conv ... MFMA inner GEMM").

Oracle: the fp32 CPU oracle run on the SAME bf16-rounded inputs and weights.  The HIP kernel multiplies exact bf16
products and accumulates in fp32 (v_mfma_f32_32x32x16_bf16), so the only differences are the summation order and the
one rounding of the result to bf16 (relative 2^-9): tolerance 1e-2 of the tensor's largest magnitude per op, stated in
each assert; a chain of ~20 layers with BatchNorm between them is compared at 5e-2."""
import numpy as np
import pytest
import torch

import oracle
from tests.helpers import OracleFPN, small_scene, sort_by_loc

pytestmark = pytest.mark.gpu
TOL = 1e-2


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def bf16_round(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16).float().numpy()


def _input_bf16(dev, coords, feats, size, width):
    """sparse tensor with bf16 rows of `width` channels (zero padded) + the same values as fp32 numpy"""
    from detection_3d_amd import sparseconvnet as scn
    t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
    f = torch.nn.functional.pad(t.features, (0, width - t.features.shape[1])).to(torch.bfloat16)
    t.features = f
    return t, f.float().cpu().numpy()


@pytest.fixture(params=[1, 2, 4])
def row_blocks(request):
    """the three weight-sharing widths of k_conv_bf16, forced on for every launch size"""
    from detection_3d_amd._lib import check, lib
    check(lib().d3d_conv_bf16_tuning(request.param, 0))
    yield request.param
    check(lib().d3d_conv_bf16_tuning(2, -1))


@pytest.mark.parametrize("cin,cout", [(9, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (256, 128), (128, 256),
                                      (256, 256)])
def test_conv_ops_bf16(dev, cin, cout, row_blocks):
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.sparseconvnet import SCN
    size = (64, 64, 16)
    rng = np.random.RandomState(cin * 1000 + cout)
    _, coords, _ = small_scene(4, 6000, (1.2, 1.0, 0.3), size)
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    width = SCN.stored_planes(cin, torch.bfloat16)
    t, x = _input_bf16(dev, coords, feats, size, width)
    x = x[:, :cin]
    _, loc = oracle.input_sites(coords)
    torch.manual_seed(0)
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    res = torch.randn(loc.shape[0], cout, device=dev).to(torch.bfloat16)
    got = conv(t).features
    assert got.dtype == torch.bfloat16 and got.shape == (loc.shape[0], cout)
    got_res = conv(t, residual=scn.SparseConvNetTensor(res, t.metadata, t.spatial_size)).features
    w = bf16_round(conv.weight.detach().cpu().numpy().reshape(27, cin, cout))
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    want = oracle.nbr_conv(x, w, nbr)
    assert rel_err(got.float().cpu().numpy(), want) < TOL
    assert rel_err(got_res.float().cpu().numpy(), want + res.float().cpu().numpy()) < TOL
    # strided 2/2 and its deconvolution
    down = scn.Convolution(3, cin, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
    d = down(t)
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], [32, 32, 8])
    wd = bf16_round(down.weight.detach().cpu().numpy().reshape(8, cin, cout))
    want_d = oracle.rule_conv(x, wd, ru, lo.shape[0])
    assert np.array_equal(d.get_spatial_locations().cpu().numpy(), lo.astype(np.int64))
    assert rel_err(d.features.float().cpu().numpy(), want_d) < TOL
    up = scn.Deconvolution(3, cout, 64, [2, 2, 2], [2, 2, 2], False).to(dev)
    u = up(d)
    wu = bf16_round(up.weight.detach().cpu().numpy().reshape(8, cout, 64))
    want_u = oracle.rule_conv(d.features.float().cpu().numpy(), wu, ru, loc.shape[0], deconv=True)
    assert u.features.shape == (loc.shape[0], 64) and u.features.dtype == torch.bfloat16
    assert rel_err(u.features.float().cpu().numpy(), want_u) < TOL


@pytest.mark.parametrize("c", [32, 128, 256])
def test_bn_prologue_bf16_fused_equals_unfused_and_oracle(dev, c, row_blocks):
    """BatchNorm + ReLU deferred into the bf16 gather: the same bits as the materialised bf16 tensor, and both within
    tolerance of the fp32 oracle on the bf16-rounded input; a NaN in row 0 stays in the rows that gather row 0."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.sparseconvnet import modules
    size = (64, 64, 16)
    rng = np.random.RandomState(c)
    _, coords, _ = small_scene(5, 6000, (1.2, 1.0, 0.3), size)
    feats = (rng.randn(coords.shape[0], c) * 2 + 0.5).astype(np.float32)
    t, x = _input_bf16(dev, coords, feats, size, c)
    _, loc = oracle.input_sites(coords)
    torch.manual_seed(1)
    bn = scn.BatchNormLeakyReLU(c, momentum=0.95, leakiness=0, track_running_stats=False).to(dev).eval()
    conv = scn.SubmanifoldConvolution(3, c, 64, 3, False).to(dev)
    bn.weight.uniform_(0.5, 1.5)
    bn.bias.uniform_(-0.5, 0.5)
    outs = []
    for fused in (True, False):
        modules.FUSE_BN_INTO_CONV = fused
        try:
            outs.append(conv(bn(t)).features.clone())
        finally:
            modules.FUSE_BN_INTO_CONV = True
    assert torch.equal(outs[0], outs[1])
    mean = x.mean(0, dtype=np.float64).astype(np.float32)
    var = x.var(0, ddof=1, dtype=np.float64).astype(np.float32)
    y, *_ = oracle.bn_forward(x, mean, var, bn.weight.cpu().numpy(), bn.bias.cpu().numpy(), 1e-4, 0.95, False, 0.0)
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    want = oracle.nbr_conv(bf16_round(y), bf16_round(conv.weight.cpu().numpy().reshape(27, c, 64)), nbr)
    assert rel_err(outs[0].float().cpu().numpy(), want) < 2 * TOL
    # absent neighbours are exact zeros by a select: a NaN row 0 reaches only the rows that really gather row 0
    f2 = t.features.clone()
    f2[0] = float("nan")
    got = conv(scn.SparseConvNetTensor(f2, t.metadata, t.spatial_size)).features.float()
    touched = np.unique(np.nonzero(nbr == 0)[0])
    bad = torch.nonzero(torch.isnan(got).any(1)).view(-1).cpu().numpy()
    assert np.array_equal(bad, touched)


def _mini_fpn(dev):
    from detection_3d_amd import sparseconvnet as scn
    torch.manual_seed(0)
    net = scn.FPN_Net([256, 256, 32], 3, ['xyz', 'color', 'normal'], 1, [32, 64, 64, 128, 128], nPlaneM=128,
                      residual_blocks=True, fpn_scales_from_top=[2, 1], roi_scales_from_top=(2, 1),
                      downsample=[[[2, 2, 2]] * 4] * 2, rpn_map_sizes=[[64, 64, 8], [32, 32, 4]],
                      voxel_scale=50, rpn_3d_2d_selector=[1, 2, 3], bn_momentum=0.95, track_running_stats=False)
    return net.to(dev).eval()


def test_backbone_bf16_vs_oracle_and_fp32(dev):
    size = (256, 256, 32)
    _, coords, feats = small_scene(7, 60000, (5.0, 4.0, 0.6), size)
    net = _mini_fpn(dev)
    inp = [torch.from_numpy(coords), torch.from_numpy(feats).to(dev)]
    rpn32, roi32 = net(inp)
    net.compute_dtype = torch.bfloat16
    try:
        rpn16, roi16 = net(inp)
        rpn16b, _ = net(inp)
    finally:
        net.compute_dtype = torch.float32
    orc = OracleFPN(net.state_dict(), size, 5, [2, 1], [2, 1], [1, 2, 3])
    rpn_w, roi_w = orc(coords, feats)
    assert all(t.features.dtype == torch.float32 for t in rpn16 + roi16)          # handed to the fp32 tail
    for a, b in zip(rpn16, rpn16b):
        assert torch.equal(a.features, b.features)                                # deterministic to the bit
    for got16, got32, (wf, wl) in list(zip(rpn16, rpn32, rpn_w)) + list(zip(roi16, roi32, [r[:2] for r in roi_w])):
        gl = got16.get_spatial_locations().cpu().numpy()
        g16, gl = sort_by_loc(got16.features.cpu().numpy(), gl)
        g32, _ = sort_by_loc(got32.features.cpu().numpy(), got32.get_spatial_locations().cpu().numpy())
        wf, wl = sort_by_loc(wf, wl)
        assert np.array_equal(gl, wl.astype(np.int64))
        assert rel_err(g16, wf) < 5e-2, rel_err(g16, wf)
        assert rel_err(g16, g32) < 5e-2
        assert rel_err(g32, wf) < 2e-4


@pytest.mark.parametrize("n,c,width", [(1, 9, 16), (1000, 9, 16), (4097, 3, 16), (777, 32, 32), (50, 20, 32), (0, 9, 16)])
def test_rows_to_bf16_is_pad_and_cast(dev, n, c, width):
    """d3d_rows_to_bf16 == F.pad(x, (0, width - c)).to(bfloat16) bit for bit (round to nearest even, channels past c zero,
    NaN / Inf kept): the hand-over of the input layer's fp32 means to the bf16 backbone."""
    from detection_3d_amd.sparseconvnet import SCN
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 100).to(dev)
    if n > 3:
        x[0, 0], x[1, 1], x[2, 2] = float("inf"), float("-inf"), 1.0 + 2.0 ** -9     # (a tie of the rounding)
        x[3, 0] = float("nan")
    got = SCN.rows_to_bf16(x, width)
    want = torch.nn.functional.pad(x, (0, width - c)).to(torch.bfloat16)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert torch.equal(got.view(torch.int16)[~torch.isnan(want)], want.view(torch.int16)[~torch.isnan(want)])
    assert torch.equal(torch.isnan(got), torch.isnan(want))

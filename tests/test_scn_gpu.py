"""GPU parity tests of the sparse-conv path: HIP (through the C ABI / host mirror) vs the CPU
oracle on the same seeded inputs.  Integer results (site ids, point lists, rulebooks, coarse
grids) are compared bit-exactly; fp32 features within 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

import oracle
from tests.helpers import OracleFPN, canon_rules, nbr_to_rules, small_scene, sort_by_loc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield
RTOL = 1e-4


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _input(dev, coords, feats, size):
    from detection_3d_amd import sparseconvnet as scn
    layer = scn.InputLayer(3, size, mode=4)
    return layer([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])


@pytest.mark.parametrize("n_points,extent,size", [
    (3000, (1.2, 1.0, 0.3), (64, 64, 16)),
    (40000, (5.0, 4.0, 0.6), (256, 256, 32)),
])
def test_input_layer_exact(dev, n_points, extent, size):
    _, coords, feats = small_scene(1, n_points, extent, size)
    t = _input(dev, coords, feats, size)
    sop, loc = oracle.input_sites(coords)
    got_loc = t.get_spatial_locations().cpu().numpy()
    assert got_loc.shape[0] == loc.shape[0]
    assert np.array_equal(got_loc, loc.astype(np.int64))          # first-occurrence ids, exact
    off, idx = t.metadata.export_input_rules(coords.shape[0])
    rules = oracle.input_rule_table(sop, loc.shape[0])
    off, idx = off.cpu().numpy(), idx.cpu().numpy()
    assert np.array_equal(np.diff(off), rules[:, 0])
    for r in np.random.RandomState(0).randint(0, loc.shape[0], 200):
        assert np.array_equal(idx[off[r]:off[r + 1]], rules[r, 1:1 + rules[r, 0]])
    want = oracle.input_forward(feats, sop, loc.shape[0], True)
    assert np.array_equal(t.features.cpu().numpy(), want)          # same fp32 op order -> bit exact


def test_input_layer_duplicates_and_batch(dev):
    # many points per voxel + two batch samples (ragged): ids global over samples
    rng = np.random.RandomState(3)
    c0 = rng.randint(0, 6, (500, 3))
    c1 = rng.randint(0, 6, (137, 3))
    coords = np.concatenate([np.concatenate([c0, np.zeros((500, 1), int)], 1),
                             np.concatenate([c1, np.ones((137, 1), int)], 1)]).astype(np.int64)
    feats = rng.randn(637, 9).astype(np.float32)
    t = _input(dev, coords, feats, (8, 8, 8))
    sop, loc = oracle.input_sites(coords)
    assert np.array_equal(t.get_spatial_locations().cpu().numpy(), loc.astype(np.int64))
    want = oracle.input_forward(feats, sop, loc.shape[0], True)
    assert np.array_equal(t.features.cpu().numpy(), want)


def test_rulebooks_exact(dev):
    size = (256, 256, 32)
    _, coords, feats = small_scene(2, 40000, (5.0, 4.0, 0.6), size)
    t = _input(dev, coords, feats, size)
    m = t.metadata
    from detection_3d_amd._lib import check, ints, lib, stream_of
    import ctypes
    _, loc = oracle.input_sites(coords)
    # submanifold 3x3x3 and 1x1x1
    for filt in ([3, 3, 3], [1, 1, 1]):
        nr = ctypes.c_long(0)
        check(lib().d3d_subm_prepare(m._h, ints(size), ints(filt), stream_of(), ctypes.byref(nr)))
        nbr, total = oracle.subm_nbr(loc, filt)
        assert nr.value == total
        got = canon_rules(m.export_rules(0, size, filt).cpu().numpy())
        assert np.array_equal(got, canon_rules(nbr_to_rules(nbr)))
    # strided 2/2 chain and the [1,1,Z] projection
    cur_size, cur_loc = list(size), loc
    for _ in range(3):
        out_size = [s // 2 for s in cur_size]
        n_out = ctypes.c_int(0)
        nr = ctypes.c_long(0)
        check(lib().d3d_conv_prepare(m._h, ints(cur_size), ints(out_size), ints([2, 2, 2]), ints([2, 2, 2]),
                                     stream_of(), ctypes.byref(n_out), ctypes.byref(nr)))
        lo, ru = oracle.conv_rules(cur_loc, [2, 2, 2], [2, 2, 2], out_size)
        assert n_out.value == lo.shape[0] and nr.value == ru.shape[0]
        got_loc = m.getSpatialLocations(out_size).cpu().numpy()
        assert np.array_equal(got_loc, lo.astype(np.int64))       # canonical first-touch numbering
        got = canon_rules(m.export_rules(1, cur_size, [2, 2, 2], [2, 2, 2]).cpu().numpy())
        assert np.array_equal(got, canon_rules(ru))
        cur_size, cur_loc = out_size, lo
    z = cur_size[2]
    out_size = [cur_size[0], cur_size[1], 1]
    n_out = ctypes.c_int(0)
    check(lib().d3d_conv_prepare(m._h, ints(cur_size), ints(out_size), ints([1, 1, z]), ints([1, 1, 1]),
                                 stream_of(), ctypes.byref(n_out), None))
    lo, ru = oracle.conv_rules(cur_loc, [1, 1, z], [1, 1, 1], out_size)
    assert n_out.value == lo.shape[0]
    assert np.array_equal(m.getSpatialLocations(out_size).cpu().numpy(), lo.astype(np.int64))
    got = canon_rules(m.export_rules(1, cur_size, [1, 1, z], [1, 1, 1]).cpu().numpy())
    assert np.array_equal(got, canon_rules(ru))


@pytest.mark.parametrize("cin,cout", [(9, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (256, 128), (128, 256),
                                      (256, 256)])
def test_conv_ops(dev, cin, cout):
    from detection_3d_amd import sparseconvnet as scn
    size = (64, 64, 16)
    rng = np.random.RandomState(cin * 1000 + cout)
    _, coords, _ = small_scene(4, 6000, (1.2, 1.0, 0.3), size)
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    layer = scn.InputLayer(3, size, mode=4)
    t = layer([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
    sop, loc = oracle.input_sites(coords)
    x = oracle.input_forward(feats, sop, loc.shape[0], True)
    torch.manual_seed(0)
    # submanifold 3^3 (+ fused residual)
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    res = torch.randn(loc.shape[0], cout, device=dev)
    got = conv(t).features.cpu().numpy()
    got_res = conv(t, residual=scn.SparseConvNetTensor(res, t.metadata, t.spatial_size)).features.cpu().numpy()
    w = conv.weight.detach().cpu().numpy().reshape(27, cin, cout)
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    want = oracle.nbr_conv(x, w, nbr)
    assert rel_err(got, want) < RTOL
    assert rel_err(got_res, want + res.cpu().numpy()) < RTOL
    # strided 2/2 and its deconvolution
    down = scn.Convolution(3, cin, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
    d = down(t)
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], [32, 32, 8])
    wd = down.weight.detach().cpu().numpy().reshape(8, cin, cout)
    want_d = oracle.rule_conv(x, wd, ru, lo.shape[0])
    assert np.array_equal(d.get_spatial_locations().cpu().numpy(), lo.astype(np.int64))
    assert rel_err(d.features.cpu().numpy(), want_d) < RTOL
    up = scn.Deconvolution(3, cout, cin if cin % 32 == 0 else 32, [2, 2, 2], [2, 2, 2], False).to(dev)
    u = up(d)
    wu = up.weight.detach().cpu().numpy().reshape(8, cout, -1)
    want_u = oracle.rule_conv(want_d, wu, ru, loc.shape[0], deconv=True)
    assert u.features.shape[0] == loc.shape[0]
    assert rel_err(u.features.cpu().numpy(), want_u) < RTOL


def test_batchnorm(dev):
    from detection_3d_amd import sparseconvnet as scn
    rng = np.random.RandomState(5)
    for C, rows in ((32, 5000), (64, 5000), (128, 4999), (256, 5000), (8, 37), (512, 3001), (1024, 700)):
        x = (rng.randn(rows, C) * 2 + 0.5).astype(np.float32)
        t = scn.SparseConvNetTensor(torch.from_numpy(x).to(dev), None, torch.tensor([8, 8, 8]))
        # eval, batch statistics (TRACK_RUNNING_STATS False)
        bn = scn.BatchNormLeakyReLU(C, momentum=0.95, leakiness=0, track_running_stats=False).to(dev).eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        got = bn(t).features.cpu().numpy()
        mean = x.mean(0, dtype=np.float64).astype(np.float32)
        var = x.var(0, ddof=1, dtype=np.float64).astype(np.float32)
        want, *_ = oracle.bn_forward(x, mean, var, bn.weight.detach().cpu().numpy(),
                                     bn.bias.detach().cpu().numpy(), 1e-4, 0.95, False, 0.0)
        assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
        # train mode: running statistics update (retention momentum)
        bn2 = scn.BatchNormLeakyReLU(C, momentum=0.95, leakiness=0.333).to(dev).train()
        got2 = bn2(t).features.cpu().numpy()
        want2, sm, si, rm, rv = oracle.bn_forward(x, np.zeros(C), np.ones(C), np.ones(C), np.zeros(C),
                                                  1e-4, 0.95, True, 0.333)
        assert np.abs(got2 - want2).max() < 2e-4 * max(1.0, np.abs(want2).max())
        assert np.allclose(bn2.running_mean.cpu().numpy(), rm, rtol=1e-4, atol=1e-6)
        assert np.allclose(bn2.running_var.cpu().numpy(), rv, rtol=1e-4, atol=1e-6)


def test_sparse_to_dense(dev):
    from detection_3d_amd import sparseconvnet as scn
    size = (32, 32, 8)
    _, coords, feats = small_scene(6, 2000, (0.6, 0.6, 0.15), size)
    t = _input(dev, coords, feats, size)
    dense = scn.SparseToDense(3, 9)(t, batch_size=1).cpu().numpy()
    sop, loc = oracle.input_sites(coords)
    want = oracle.sparse_to_dense(oracle.input_forward(feats, sop, loc.shape[0], True), loc, size, 1)
    assert np.array_equal(dense, want)


def _mini_fpn(dev, fuse, skip):
    from detection_3d_amd import sparseconvnet as scn
    torch.manual_seed(0)
    net = scn.FPN_Net([256, 256, 32], 3, ['xyz', 'color', 'normal'], 1, [32, 64, 64, 128, 128], nPlaneM=128,
                      residual_blocks=True, fpn_scales_from_top=[2, 1], roi_scales_from_top=(2, 1),
                      downsample=[[[2, 2, 2]] * 4] * 2, rpn_map_sizes=[[64, 64, 8], [32, 32, 4]],
                      voxel_scale=50, rpn_3d_2d_selector=[1, 2, 3], bn_momentum=0.95,
                      track_running_stats=False, fuse_adds=fuse, skip_unused=skip)
    return net.to(dev).eval()


@pytest.mark.parametrize("fuse,skip", [(True, True), (False, False)])
def test_fpn_backbone_vs_oracle(dev, fuse, skip):
    size = (256, 256, 32)
    _, coords, feats = small_scene(7, 60000, (5.0, 4.0, 0.6), size)
    net = _mini_fpn(dev, fuse, skip)
    rpn, roi = net([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
    orc = OracleFPN(net.state_dict(), size, 5, [2, 1], [2, 1], [1, 2, 3])
    rpn_w, roi_w = orc(coords, feats)
    for got, (wf, wl) in list(zip(rpn, rpn_w)) + list(zip(roi, [r[:2] for r in roi_w])):
        gl = got.get_spatial_locations().cpu().numpy()
        gf, gl = sort_by_loc(got.features.cpu().numpy(), gl)
        wf, wl = sort_by_loc(wf, wl)
        assert np.array_equal(gl, wl.astype(np.int64))
        assert rel_err(gf, wf) < 2e-4, rel_err(gf, wf)


def test_bn_prologue_fusion_is_bit_identical(dev):
    """Inference BatchNorm(+leaky ReLU) deferred into the consuming convolution's gather (d3d_bn_prologue)
    must give exactly the tensors of the unfused path: same float expressions, no contraction."""
    from detection_3d_amd.sparseconvnet import modules
    size = (256, 256, 32)
    _, coords, feats = small_scene(11, 50000, (5.0, 4.0, 0.6), size)
    net = _mini_fpn(dev, True, True)
    outs = []
    for fused in (True, False):
        modules.FUSE_BN_INTO_CONV = fused
        try:
            rpn, roi = net([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
        finally:
            modules.FUSE_BN_INTO_CONV = True
        outs.append([t.features.clone() for t in list(rpn) + list(roi)])
    assert len(outs[0]) == len(outs[1]) > 0
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_sites", [1, 31, 32, 33, 257, 8191, 8192, 8193, 20000])
def test_plan_and_conv_at_boundary_sizes(dev, n_sites):
    """Row counts around the block (32), small-plan (8192) and offset-split thresholds: rulebooks exact, fused
    BatchNorm + convolution + residual within tolerance."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd._lib import check, ints, lib, stream_of
    import ctypes
    size = (64, 64, 32)
    rng = np.random.RandomState(n_sites)
    # distinct voxels in a compact region (dense enough to have neighbours)
    side = max(2, int(np.ceil((n_sites * 2.5) ** (1 / 3))))
    cells = rng.permutation(side ** 3)[:n_sites]
    coords = np.stack([cells // (side * side), (cells // side) % side, cells % side], 1).astype(np.int64)
    coords = coords[(coords < np.array(size)).all(1)]
    cin = cout = 64
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
    sop, loc = oracle.input_sites(coords)
    assert t.features.shape[0] == loc.shape[0] == coords.shape[0]
    x = oracle.input_forward(feats, sop, loc.shape[0], True)
    nr = ctypes.c_long(0)
    check(lib().d3d_subm_prepare(t.metadata._h, ints(size), ints([3, 3, 3]), stream_of(), ctypes.byref(nr)))
    nbr, total = oracle.subm_nbr(loc, [3, 3, 3])
    assert nr.value == total
    assert np.array_equal(canon_rules(t.metadata.export_rules(0, size, [3, 3, 3]).cpu().numpy()),
                          canon_rules(nbr_to_rules(nbr)))
    torch.manual_seed(1)
    bn = scn.BatchNormLeakyReLU(cin, momentum=0.95, leakiness=0, track_running_stats=False).to(dev).eval()
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        res = torch.randn(loc.shape[0], cout, device=dev)
        got = conv(bn(t), residual=scn.SparseConvNetTensor(res, t.metadata, t.spatial_size)).features.cpu().numpy()
    if loc.shape[0] > 1:
        mean = x.mean(0, dtype=np.float64).astype(np.float32)
        var = x.var(0, ddof=1, dtype=np.float64).astype(np.float32)
        y, *_ = oracle.bn_forward(x, mean, var, bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(), 1e-4,
                                  0.95, False, 0.0)
        want = oracle.nbr_conv(y, conv.weight.detach().cpu().numpy().reshape(27, cin, cout), nbr) + res.cpu().numpy()
        assert rel_err(got, want) < 2e-4
    else:
        assert got.shape == (1, cout)          # one site: the unbiased variance is undefined (reference: NaN)


def test_one_parity_sites_fill_their_table_class_and_still_terminate(dev):
    """Every voxel on even coordinates: all keys fall into ONE of the eight position classes of the locality-preserving
    hash, i.e. 4x more keys than that class has slots.  The probe sequence moves on to the other classes
    (probe_next), so the input layer, the submanifold probes and a strided grid still terminate and match the oracle."""
    import ctypes
    from detection_3d_amd._lib import check, ints, lib, stream_of
    size = (64, 64, 16)
    rng = np.random.RandomState(4)
    cells = rng.permutation(32 * 32 * 8)[:3000]
    coords = np.stack([2 * (cells // 256), 2 * ((cells // 8) % 32), 2 * (cells % 8), np.zeros_like(cells)], 1).astype(np.int64)
    feats = rng.randn(coords.shape[0], 9).astype(np.float32)
    t = _input(dev, coords, feats, size)
    sop, loc = oracle.input_sites(coords)
    assert np.array_equal(t.get_spatial_locations().cpu().numpy(), loc.astype(np.int64))
    m = t.metadata
    nr = ctypes.c_long(0)
    check(lib().d3d_subm_prepare(m._h, ints(size), ints([3, 3, 3]), stream_of(), ctypes.byref(nr)))
    nbr, total = oracle.subm_nbr(loc, [3, 3, 3])
    assert nr.value == total == loc.shape[0]                 # even coordinates: no two sites are neighbours
    out_size = [32, 32, 8]
    n_out = ctypes.c_int(0)
    check(lib().d3d_conv_prepare(m._h, ints(size), ints(out_size), ints([2, 2, 2]), ints([2, 2, 2]), stream_of(),
                                 ctypes.byref(n_out), None))
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], out_size)
    assert n_out.value == lo.shape[0] == 3000
    assert np.array_equal(m.getSpatialLocations(out_size).cpu().numpy(), lo.astype(np.int64))
    got = canon_rules(m.export_rules(1, size, [2, 2, 2], [2, 2, 2]).cpu().numpy())
    assert np.array_equal(got, canon_rules(ru))


@pytest.mark.parametrize("fused", [False, True])
def test_non_finite_row_stays_local(dev, fused):
    """An absent neighbour contributes exact zeros that never pass through a real row's values: a NaN in feature row 0
    (the row that used to stand in for absent neighbours) reaches only the outputs that really gather row 0 -- as in
    the reference, which skips absent rules (SCN/CPU/Convolution.cpp rule loops)."""
    from detection_3d_amd import sparseconvnet as scn
    size = (64, 64, 16)
    rng = np.random.RandomState(9)
    _, coords, _ = small_scene(4, 6000, (1.2, 1.0, 0.3), size)
    feats = rng.randn(coords.shape[0], 64).astype(np.float32)
    t = _input(dev, coords, feats, size)
    _, loc = oracle.input_sites(coords)
    torch.manual_seed(2)
    conv = scn.SubmanifoldConvolution(3, 64, 64, 3, False).to(dev)
    f2 = t.features.clone()
    f2[0] = float("nan")
    x = scn.SparseConvNetTensor(f2, t.metadata, t.spatial_size)
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    touched = np.unique(np.nonzero(nbr == 0)[0])
    if fused:    # deferred BatchNorm with given statistics (a NaN row would poison batch statistics for every row)
        mean, invstd = torch.zeros(64, device=dev), torch.ones(64, device=dev)
        from detection_3d_amd.sparseconvnet.modules import _PendingBN
        # leaky activation: with leakiness 0 the ReLU's max(t, 0) already turns a NaN into 0 (bn.hip)
        x = _PendingBN(f2, (mean, invstd, torch.ones(64, device=dev), torch.full((64,), 0.25, device=dev), 0.333),
                       t.metadata, t.spatial_size)
    got = conv(x).features
    bad = torch.nonzero(torch.isnan(got).any(1)).view(-1).cpu().numpy()
    assert np.array_equal(bad, touched) and 0 < len(touched) < 60


@pytest.mark.parametrize("n_points,cin,cout", [(40000, 32, 32), (40000, 64, 64), (40000, 64, 128), (3000, 128, 128),
                                               (600, 256, 256), (40, 128, 256)])
def test_conv_epilogue_column_sums_give_the_batchnorm_statistics(dev, n_points, cin, cout):
    """d3d_bn_prologue.out_stats: the per-row-block fp64 column sums a convolution leaves (unsplit launches: from k_conv's
    epilogue; offset-split few-row launches: from k_conv_reduce), finished by d3d_bn_stats_from_partials, are the
    statistics d3d_bn_batch_invstd / d3d_bn_batch_stats compute from the stored tensor -- fp64 sums in another order, so
    equal after the rounding to fp32 up to 1 ulp: tolerance 5e-7 relative.  Submanifold (+ residual), strided, deconvolution."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.sparseconvnet import SCN, modules
    size = (128, 128, 32)
    _, coords, _ = small_scene(n_points, n_points, (2.5, 2.0, 0.6), size)
    rng = np.random.RandomState(cin + cout)
    feats = torch.from_numpy((rng.randn(coords.shape[0], cin) * 3 + 1).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    with torch.no_grad():
        t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), feats])
        sub = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
        down = scn.Convolution(3, cin, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
        up = scn.Deconvolution(3, cout, cout, [2, 2, 2], [2, 2, 2], False).to(dev)
        res = scn.SparseConvNetTensor(torch.randn(t.features.shape[0], cout, device=dev), t.metadata, t.spatial_size)
        outs = [sub(t), sub(t, residual=res), down(t)]
        outs.append(up(outs[2], residual=res))
        for o in outs:
            cp = modules._col_partials(o, o.features)
            assert cp is not None and cp[1] > 0
            f = o.features
            for want_invstd, direct in ((True, SCN.batch_mean_invstd(f, 1e-4)), (False, SCN.batch_stats(f))):
                got = SCN.stats_from_partials(cp[0], cp[1], f.shape[0], 1e-4, want_invstd=want_invstd)
                for g, w in zip(got, direct):
                    assert torch.isfinite(g).all()
                    assert ((g - w).abs() <= 5e-7 * w.abs().clamp(min=1e-3)).all(), (g - w).abs().max()
            x = f.double()
            assert ((got[0].double() - x.mean(0)).abs() <= 1e-6 * x.abs().mean(0).clamp(min=1e-3)).all()
        # a tensor that was modified or replaced after the convolution wrote it no longer matches its partials
        o = outs[0]
        o.features.add_(1.0)
        assert modules._col_partials(o, o.features) is None


def test_backbone_with_and_without_epilogue_statistics(dev):
    """FUSE_BN_STATS only changes where the BatchNorm sums come from: every output map within 1e-5 of the pass that
    re-reads the tensors (statistics equal to 1 ulp, amplified through ~20 layers), and deterministic to the bit."""
    from detection_3d_amd.sparseconvnet import modules
    size = (256, 256, 32)
    _, coords, feats = small_scene(12, 50000, (5.0, 4.0, 0.6), size)
    net = _mini_fpn(dev, True, True)
    outs = []
    for fused in (True, True, False):
        modules.FUSE_BN_STATS = fused
        try:
            rpn, roi = net([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
        finally:
            modules.FUSE_BN_STATS = True
        outs.append([t.features.clone() for t in list(rpn) + list(roi)])
    for a, b, c in zip(*outs):
        assert torch.equal(a, b)
        assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1e-5


def test_geometry_chain_on_the_library_thread_equals_the_calls_made_in_line(dev):
    """d3d_geometry_async_start / _wait / _finish: the chain of strided grids built by the library's own thread gives the
    grids (site counts, coordinates in the same numbering) and rulebooks of the same d3d_conv_prepare calls made one
    after the other by the caller; an impossible entry is reported by wait / finish instead of hanging."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd._lib import D3DError
    size = (256, 256, 32)
    _, coords, feats = small_scene(21, 60000, (5.0, 4.0, 0.6), size)
    inp = [torch.from_numpy(coords), torch.from_numpy(feats).to(dev)]
    specs, cur = [], list(size)
    for _ in range(4):
        nxt = [v // 2 for v in cur]
        specs.append([1] + cur + nxt + [2, 2, 2] + [2, 2, 2])
        cur = nxt
    specs.append([1] + cur + [cur[0], cur[1], 1] + [1, 1, cur[2]] + [1, 1, 1])     # a z-collapsing projection grid
    with torch.no_grad():
        a = scn.InputLayer(3, size, mode=4)(inp)
        b = scn.InputLayer(3, size, mode=4)(inp)
    want = [scn.SCN.Convolution_prepare(sp[1:4], sp[4:7], sp[7:10], sp[10:13], a.metadata) for sp in specs]
    # views of the grids, listed behind them: 3x3x3 submanifold rulebooks (kind 0) and deconvolution views (kind 2)
    views = [[0] + sp[1:4] + sp[1:4] + [3, 3, 3] + [1, 1, 1] for sp in specs[:4]]
    views += [[2] + sp[4:7] + sp[1:4] + sp[7:10] + sp[10:13] for sp in specs[:4]]
    for v in views:
        if v[0] == 0:
            scn.SCN.SubmanifoldConvolution_prepare(v[1:4], v[7:10], a.metadata)
        else:
            scn.SCN.Deconvolution_prepare(v[1:4], v[4:7], v[7:10], v[10:13], a.metadata)
    main = torch.cuda.current_stream(dev)
    geo, plan = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    md = b.metadata
    geo.wait_stream(main)
    plan.wait_stream(main)
    md.set_geometry_stream(geo.cuda_stream)
    md.set_plan_stream(plan.cuda_stream)
    try:
        with pytest.raises(Exception, match="plan stream"):
            md.geometry_async_start(specs + views, geo.cuda_stream)                     # views need their stream
        md.geometry_async_start(specs + views, geo.cuda_stream, plan.cuda_stream)
        got = [md.geometry_async_wait(i, main.cuda_stream) for i in (2, 0, 4, 1, 3)]     # any order, any number of times
        md.geometry_async_wait(len(specs) + len(views) - 1, main.cuda_stream)
        md.geometry_async_finish()
    finally:
        main.wait_stream(geo)
        main.wait_stream(plan)
        md.set_plan_stream(None)
        md.set_geometry_stream(None)
    assert got == [want[i] for i in (2, 0, 4, 1, 3)] and min(want) > 0
    for v in views:
        ra = a.metadata.export_rules(v[0], v[4:7] if v[0] == 2 else v[1:4], v[7:10], v[10:13] if v[0] == 2 else None)
        rb = md.export_rules(v[0], v[4:7] if v[0] == 2 else v[1:4], v[7:10], v[10:13] if v[0] == 2 else None)
        assert ra.shape[0] > 0 and np.array_equal(canon_rules(ra.cpu().numpy()), canon_rules(rb.cpu().numpy()))
    for sp in specs:
        la, lb = a.metadata.getSpatialLocations(sp[4:7]), md.getSpatialLocations(sp[4:7])
        assert torch.equal(la, lb)
        ra = a.metadata.export_rules(1, sp[1:4], sp[7:10], sp[10:13])
        rb = md.export_rules(1, sp[1:4], sp[7:10], sp[10:13])
        assert np.array_equal(canon_rules(ra.cpu().numpy()), canon_rules(rb.cpu().numpy()))
    # an entry that cannot be built (its input grid does not exist): error on wait and on finish, no hang
    with torch.no_grad():
        c = scn.InputLayer(3, size, mode=4)(inp)
    md = c.metadata
    geo.wait_stream(main)
    md.set_geometry_stream(geo.cuda_stream)
    try:
        md.geometry_async_start([specs[0], [1, 64, 64, 8, 32, 32, 4, 2, 2, 2, 2, 2, 2]], geo.cuda_stream)
        assert md.geometry_async_wait(0, main.cuda_stream) == want[0]
        with pytest.raises(D3DError, match="geometry thread"):
            md.geometry_async_wait(1, main.cuda_stream)
        with pytest.raises(D3DError, match="geometry thread"):
            md.geometry_async_finish()
    finally:
        main.wait_stream(geo)
        md.set_geometry_stream(None)


def test_conv_variants_are_bit_identical(dev):
    """The measured variants of the fp32 convolution produce the same bits as k_conv's default form: gathers requested ahead
    of a step's matrix work (d3d_conv_late_mode 0) and the weight-sharing kernel of conv_ws.hip (d3d_conv_ws_mode 2: every
    64 -> 64 / 128 -> 128 launch) -- with the fused BatchNorm prologue, the residual and the statistics epilogue."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd._lib import lib
    size = (128, 128, 32)
    _, coords, _ = small_scene(21, 40000, (2.5, 2.5, 0.6), size)
    rng = np.random.RandomState(5)
    for c in (64, 128):
        feats = rng.randn(coords.shape[0], c).astype(np.float32)
        outs = []
        for late, ws in ((1, 0), (0, 0), (1, 2)):
            lib().d3d_conv_late_mode(late)
            lib().d3d_conv_ws_mode(ws)
            try:
                with torch.no_grad():
                    t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
                    torch.manual_seed(c)
                    bn = scn.BatchNormLeakyReLU(c, momentum=0.95, leakiness=0, track_running_stats=False).to(dev).eval()
                    conv = scn.SubmanifoldConvolution(3, c, c, 3, False).to(dev)
                    y = conv(bn(t), residual=t)
                    y2 = conv(y)
                    outs.append((y.features.clone(), y2.features.clone()))
            finally:
                lib().d3d_conv_late_mode(1)
                lib().d3d_conv_ws_mode(0)
        assert outs[0][0].abs().sum() > 0
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]), c

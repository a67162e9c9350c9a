"""CPU tests pinning the sparse-conv oracle against independent known answers: torch's dense
conv3d / conv_transpose3d on densified grids, numpy group-bys, and analytic rule counts.
(The reference's C++ cannot be built here -- sparsehash is absent -- and ships no fixtures.)"""
import numpy as np
import torch
import torch.nn.functional as F

import oracle
from tests.helpers import small_scene


def _dense(feats, loc, size):
    d = np.zeros((1, feats.shape[1]) + tuple(size), np.float32)
    d[0][:, loc[:, 0], loc[:, 1], loc[:, 2]] = feats.T
    return torch.from_numpy(d)


def test_voxelize_matches_formula():
    pcl, coords, feats = small_scene(0, 5000, (1.2, 1.0, 0.3), (64, 64, 16))
    a = pcl[:, :3].astype(np.float64) * 50.0
    a -= a.min(0)
    assert np.array_equal(coords, np.trunc(a).astype(np.int64))
    assert np.array_equal(feats[:, :3], (a / 50.0).astype(np.float32))
    assert np.array_equal(feats[:, 3:], pcl[:, 3:])
    # points beyond full_scale are dropped (suncg_dataset.py:160-172)
    c2, f2 = oracle.voxelize(pcl, 50, (32, 64, 16))
    assert c2.shape[0] < coords.shape[0] and c2[:, 0].max() < 32


def test_input_layer_first_occurrence_and_mean():
    rng = np.random.RandomState(1)
    coords = rng.randint(0, 5, (400, 3)).astype(np.int64)
    feats = rng.randn(400, 4).astype(np.float32)
    sop, loc = oracle.input_sites(coords)
    seen = {}
    for i, c in enumerate(map(tuple, coords)):
        if c not in seen:
            seen[c] = len(seen)
        assert sop[i] == seen[c]
    assert np.array_equal(loc[:, :3], np.array(list(seen.keys())))
    out = oracle.input_forward(feats, sop, len(seen), True)
    for s in range(len(seen)):
        assert np.allclose(out[s], feats[sop == s].mean(0), atol=1e-6)
    rules = oracle.input_rule_table(sop, len(seen))
    assert rules.shape[1] == 1 + np.bincount(sop).max()
    assert np.array_equal(rules[:, 0], np.bincount(sop))
    assert all(np.all(np.diff(r[1:1 + r[0]]) > 0) for r in rules)      # input order


def test_submanifold_conv_equals_masked_dense_conv3d():
    size = (12, 10, 6)
    rng = np.random.RandomState(2)
    coords = np.unique(rng.randint(0, [12, 10, 6], (300, 3)), axis=0).astype(np.int64)
    rng.shuffle(coords)
    cin, cout = 5, 7
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    sop, loc = oracle.input_sites(coords)
    W = rng.randn(27, cin, cout).astype(np.float32)
    nbr, total = oracle.subm_nbr(loc, [3, 3, 3])
    got = oracle.nbr_conv(feats, W, nbr)
    # offsets enumerate (dx,dy,dz) with z fastest (RectangularRegions.h:31-38); in = out - 1 + d
    w = torch.from_numpy(W.reshape(3, 3, 3, cin, cout)).permute(4, 3, 0, 1, 2).contiguous()
    dense = F.conv3d(_dense(feats, loc, size), w, padding=1)[0].numpy()
    want = dense[:, loc[:, 0], loc[:, 1], loc[:, 2]].T
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4)
    # rule count = number of ordered active neighbour pairs incl. self
    occ = np.zeros(size, bool)
    occ[tuple(coords.T)] = True
    cnt = F.conv3d(torch.from_numpy(occ[None, None].astype(np.float32)), torch.ones(1, 1, 3, 3, 3), padding=1)[0, 0].numpy()
    assert total == int(cnt[occ].sum())


def test_strided_conv_and_deconv_equal_dense():
    size = (12, 8, 6)
    rng = np.random.RandomState(3)
    coords = np.unique(rng.randint(0, [12, 8, 6], (200, 3)), axis=0).astype(np.int64)
    rng.shuffle(coords)
    cin, cout = 4, 6
    feats = rng.randn(coords.shape[0], cin).astype(np.float32)
    _, loc = oracle.input_sites(coords)
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], [6, 4, 3])
    assert ru.shape[0] == loc.shape[0]                       # k = s = 2: one rule per input site
    assert np.array_equal(np.unique(loc[:, :3] // 2, axis=0), np.unique(lo[:, :3], axis=0))
    first = {}
    for p in map(tuple, loc[:, :3] // 2):                    # first-touch numbering in input id order
        first.setdefault(p, len(first))
    assert np.array_equal(lo[:, :3], np.array(list(first.keys())))
    W = rng.randn(8, cin, cout).astype(np.float32)
    got = oracle.rule_conv(feats, W, ru, lo.shape[0])
    w = torch.from_numpy(W.reshape(2, 2, 2, cin, cout)).permute(4, 3, 0, 1, 2).contiguous()
    dense = F.conv3d(_dense(feats, loc, size), w, stride=2)[0].numpy()
    assert np.allclose(got, dense[:, lo[:, 0], lo[:, 1], lo[:, 2]].T, rtol=1e-4, atol=1e-4)
    # deconvolution = transposed conv restricted to the active fine sites (Deconvolution.cpp:17,33-37)
    Wd = rng.randn(8, cout, cin).astype(np.float32)
    back = oracle.rule_conv(got, Wd, ru, loc.shape[0], deconv=True)
    wt = torch.from_numpy(Wd.reshape(2, 2, 2, cout, cin)).permute(3, 4, 0, 1, 2).contiguous()
    dense_t = F.conv_transpose3d(_dense(got, lo, (6, 4, 3)), wt, stride=2)[0].numpy()
    assert np.allclose(back, dense_t[:, loc[:, 0], loc[:, 1], loc[:, 2]].T, rtol=1e-4, atol=1e-4)


def test_z_projection_rules():
    rng = np.random.RandomState(4)
    coords = np.unique(rng.randint(0, [6, 6, 8], (150, 3)), axis=0).astype(np.int64)
    _, loc = oracle.input_sites(coords)
    lo, ru = oracle.conv_rules(loc, [1, 1, 8], [1, 1, 1], [6, 6, 1])
    assert np.all(lo[:, 2] == 0)
    assert np.array_equal(ru[:, 2], loc[ru[:, 0], 2])        # offset = z of the input site
    assert lo.shape[0] == np.unique(loc[:, :2], axis=0).shape[0]


def test_batchnorm_matches_torch():
    rng = np.random.RandomState(5)
    x = (rng.randn(300, 8) * 3 + 1).astype(np.float32)
    w, b = rng.rand(8).astype(np.float32) + 0.5, rng.randn(8).astype(np.float32)
    out, sm, si, rm, rv = oracle.bn_forward(x, np.zeros(8), np.ones(8), w, b, 1e-4, 0.95, True, 0.0)
    ref = F.relu(F.batch_norm(torch.from_numpy(x), None, None, torch.from_numpy(w), torch.from_numpy(b), True, 0.0, 1e-4))
    assert np.allclose(out, ref.numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(rm, 0.05 * x.mean(0), rtol=1e-4, atol=1e-6)           # retention momentum
    assert np.allclose(rv, 0.95 + 0.05 * x.var(0, ddof=1), rtol=1e-4)
    out2, *_ = oracle.bn_forward(x, x.mean(0), x.var(0, ddof=1), w, b, 1e-4, 0.95, False, 0.333)
    y = (x - x.mean(0)) / np.sqrt(x.var(0, ddof=1) + 1e-4) * w + b
    assert np.allclose(out2, np.where(y > 0, y, 0.333 * y), rtol=1e-4, atol=1e-4)


def test_sparse_to_dense_layout():
    loc = np.array([[1, 2, 3, 0], [0, 0, 0, 0], [2, 1, 0, 1]], np.int32)
    f = np.arange(6, dtype=np.float32).reshape(3, 2)
    d = oracle.sparse_to_dense(f, loc, (3, 3, 4), batch=2)
    assert d.shape == (2, 2, 3, 3, 4) and d[0, 1, 1, 2, 3] == 1 and d[1, 0, 2, 1, 0] == 4
    assert np.count_nonzero(d) == 5

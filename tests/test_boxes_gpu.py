"""GPU parity of the box ops vs the oracle.  IoU: the HIP kernels follow the same arithmetic
contract (fp32 geometry, fp64 fan sum, -ffp-contract=off) and are compared bit-exactly where both
sides are IEEE basic ops, else within 1e-4 relative (BASELINE.json).  NMS survivor sets: exact."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from detection_3d_amd.synthetic import make_boxes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _close(got, want, tol=1e-4):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2, 3])
def test_rotate_iou_eval(dev, criterion):
    from detection_3d_amd import box_ops
    b, _ = make_boxes(0, 300)
    q, _ = make_boxes(1, 170)
    b2, q2 = b[:, [0, 1, 3, 4, 6]], q[:, [0, 1, 3, 4, 6]]
    q2[:17] = b2[:17]                                            # identical pairs -> forced to 1
    got = box_ops.rotate_iou_gpu_eval(torch.from_numpy(b2).to(dev), torch.from_numpy(q2).to(dev), criterion).cpu().numpy()
    want = oracle.rotate_iou_eval(b2, q2, criterion)
    assert got.shape == (300, 170)
    assert np.mean(got != want) < 1e-3 and _close(got, want)       # bit-exact but for libm last-ulp cases
    if criterion == -1:
        assert np.all(np.diag(got)[:17] == 1.0)


def test_empty_and_ragged_shapes(dev):
    from detection_3d_amd import box_ops
    e = torch.zeros((0, 5), device=dev)
    b = torch.from_numpy(make_boxes(0, 65)[0][:, [0, 1, 3, 4, 6]]).to(dev)
    assert box_ops.rotate_iou_gpu_eval(e, b).shape == (0, 65)
    assert box_ops.rotate_iou_gpu_eval(b, e).shape == (65, 0)
    assert box_ops.rotate_nms_3d(torch.zeros((0, 7), device=dev), torch.zeros(0, device=dev), 2000, 100, 0.5).numel() == 0
    one = torch.from_numpy(make_boxes(0, 1)[0]).to(dev)
    assert box_ops.rotate_nms_3d(one, torch.ones(1, device=dev), 2000, 100, 0.5).tolist() == [0]


def test_boxes_iou_3d_flags(dev):
    from detection_3d_amd import box_ops
    t, _ = make_boxes(2, 97)
    a, _ = make_boxes(3, 1030)
    tt, aa = torch.from_numpy(t).to(dev), torch.from_numpy(a).to(dev)
    for flag, aug, crit in [("rpn_post", None, -1),
                            ("rpn_label_generation", dict(target_Y=0.3, target_Z=0.25, anchor_Y=0.0, anchor_Z=0.0), 2),
                            ("roi_label_generation", dict(target_Y=0.3, target_Z=0.3, anchor_Y=0.3, anchor_Z=0.3), -1),
                            ("eval", dict(target_Y=0.2, target_Z=0.2, anchor_Y=0.2, anchor_Z=0.2), -1)]:
        got = box_ops.boxes_iou_3d(tt, aa, aug, crit, flag=flag).cpu().numpy()
        want = oracle.boxes_iou_3d(t, a, aug, crit)
        assert np.mean(got != want) < 1e-3 and _close(got, want), flag
    with pytest.raises(NotImplementedError):
        box_ops.boxes_iou_3d(tt, aa, None, -1, flag="bogus")


@pytest.mark.parametrize("n,thr,aug", [(2000, 0.5, (0.3, 0.3)), (1000, 0.45, (0.2, 0.2)), (333, 0.3, (0, 0))])
def test_nms_survivor_sets_exact(dev, n, thr, aug):
    from detection_3d_amd import box_ops
    b, s = make_boxes(10 + n, n)
    s[::7] = s[3]                                                # score ties -> lower index first
    keep = box_ops.nms_3d_clamped(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), thr, list(aug),
                                  max_proposals=1000 if n == 2000 else -1,
                                  flag="rpn_post" if n == 2000 else "roi_post").cpu().numpy()
    bc = b.copy()
    bc[:, 3:5] = np.maximum(bc[:, 3:5], aug[0])
    bc[:, 5] = np.maximum(bc[:, 5], aug[1])
    want = oracle.rotate_nms_3d(bc, s, thr)[:1000 if n == 2000 else 500]
    assert np.array_equal(keep, want)
    assert len(keep) < n                                         # something was suppressed


def test_nms_on_reference_demo_detections(dev):
    from detection_3d_amd import box_ops
    r = np.load(os.path.join(GOLD, "rooms.npz"))
    for k in sorted(r.files):
        s = r[k]
        b = s[:, [0, 1, 2, 4, 3, 5, 6]].copy()
        b[:, 2] -= s[:, 5] * 0.5
        b[:, 6] -= math.pi * 0.5
        b[:, 6] = b[:, 6] - np.floor(b[:, 6] / math.pi + 0.5) * math.pi
        b = b.astype(np.float32)
        for c in np.unique(s[:, 7]):
            bc = b[s[:, 7] == c]
            sc = np.linspace(1, 0.5, bc.shape[0]).astype(np.float32)
            got = box_ops.nms_3d_clamped(torch.from_numpy(bc).to(dev), torch.from_numpy(sc).to(dev), 0.45,
                                         [0.2, 0.2], -1, flag="roi_post").cpu().numpy()
            bb = bc.copy()
            bb[:, 3:5] = np.maximum(bb[:, 3:5], 0.2)
            bb[:, 5] = np.maximum(bb[:, 5], 0.2)
            assert np.array_equal(got, oracle.rotate_nms_3d(bb, sc, 0.45))


def test_box_decode(dev):
    from detection_3d_amd import box_ops
    g = np.load(os.path.join(GOLD, "ref_python.npz"))
    rng = np.random.RandomState(0)
    anchors = np.concatenate([rng.rand(4096, 3) * 20, 0.2 + rng.rand(4096, 3) * 3, (rng.rand(4096, 1) - 0.5) * 3], 1).astype(np.float32)
    enc = (rng.randn(4096, 7) * 0.3).astype(np.float32)
    enc[:, 6] = g["lp_in"]
    got = box_ops.box_decode(torch.from_numpy(enc).to(dev), torch.from_numpy(anchors).to(dev)).cpu().numpy()
    want = oracle.box_decode(enc, anchors)
    assert np.array_equal(got, want)                             # IEEE basic ops + sqrt + floor only
    # multi-class layout [n, 7*nc] (box_coder_3d.py:50-63)
    enc3 = (rng.randn(100, 21) * 0.3).astype(np.float32)
    got3 = box_ops.box_decode(torch.from_numpy(enc3).to(dev), torch.from_numpy(anchors[:100]).to(dev)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(got3[:, 7 * c:7 * c + 7], oracle.box_decode(enc3[:, 7 * c:7 * c + 7], anchors[:100]))
    # yaw wrap pinned to the reference's own limit_period output
    z = np.zeros((4096, 7), np.float32)
    z[:, 6] = g["lp_in"]
    a1 = np.zeros((4096, 7), np.float32)
    a1[:, 3:6] = 1
    y = box_ops.box_decode(torch.from_numpy(z).to(dev), torch.from_numpy(a1).to(dev)).cpu().numpy()
    assert np.array_equal(y[:, 6], g["lp_half"])


def test_roi_align_dense_and_sparse(dev):
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd.roi_align_rotated_3d import roi_align_rotated_3d_forward, roi_align_rotated_3d_sparse
    from tests.helpers import small_scene
    size = (64, 64, 16)
    _, coords, _ = small_scene(8, 9000, (1.2, 1.0, 0.3), size)
    rng = np.random.RandomState(1)
    C = 128
    feats = rng.randn(coords.shape[0], C).astype(np.float32)
    t = scn.InputLayer(3, size, mode=4)([torch.from_numpy(coords), torch.from_numpy(feats).to(dev)])
    sop, loc = oracle.input_sites(coords)
    f = oracle.input_forward(feats, sop, loc.shape[0], True)
    crop = (loc[:, :3].max(0) + 1).tolist()
    dense = oracle.sparse_to_dense(f, loc, size, 1)[:, :, :crop[0], :crop[1], :crop[2]].copy()
    K = 60
    rois = np.zeros((K, 8), np.float32)
    rois[:, 1] = rng.rand(K) * crop[1] * 8
    rois[:, 2] = rng.rand(K) * crop[0] * 8
    rois[:, 3] = rng.rand(K) * crop[2] * 8
    rois[:, 4] = 4 + rng.rand(K) * 200
    rois[:, 5] = 2 + rng.rand(K) * 40
    rois[:, 6] = 4 + rng.rand(K) * 100
    rois[:, 7] = rng.rand(K) * 180
    rois[:5, 1:4] += 500                                         # partly / fully outside the map
    want = oracle.roi_align_rotated_3d(dense, rois, 1.0 / 8, 6, 8, 4, 2)
    r = torch.from_numpy(rois).to(dev)
    got_d = roi_align_rotated_3d_forward(torch.from_numpy(dense).to(dev), r, 1.0 / 8, 6, 8, 4, 2).cpu().numpy()
    got_s = roi_align_rotated_3d_sparse(t, r, 1.0 / 8, 6, 8, 4, 2).cpu().numpy()
    assert got_d.shape == (K, C, 6, 8, 4)
    assert _close(got_d, want, 1e-5)
    assert _close(got_s, want, 1e-5)
    assert np.abs(want).max() > 0.1
    # inference forms: channels-inner layout and per-level selection write the same numbers elsewhere
    from detection_3d_amd.roi_align_rotated_3d import roi_align_rotated_3d_sparse_into
    gs = torch.from_numpy(got_s).to(dev)
    inner = torch.full((K, 6, 8, C, 4), 7.0, device=dev)
    roi_align_rotated_3d_sparse_into(inner, t, r, 1.0 / 8, 2, channels_inner=True)
    assert torch.equal(inner.permute(0, 3, 1, 2, 4), gs)
    levels = torch.from_numpy((np.arange(K) % 3 == 0).astype(np.int32)).to(dev)
    part = torch.full((K, C, 6, 8, 4), 7.0, device=dev)
    roi_align_rotated_3d_sparse_into(part, t, r, 1.0 / 8, 2, roi_levels=levels, level=1, channels_inner=False)
    sel = levels.bool()
    assert torch.equal(part[sel], gs[sel]) and bool((part[~sel] == 7.0).all())
    # odd channel count and adaptive sampling (sampling_ratio 0)
    t37 = scn.SparseConvNetTensor(t.features[:, :37].contiguous(), t.metadata, t.spatial_size)
    want37 = oracle.roi_align_rotated_3d(dense[:, :37].copy(), rois[:12], 1.0 / 8, 3, 2, 5, 0)
    got37 = roi_align_rotated_3d_sparse(t37, r[:12].contiguous(), 1.0 / 8, 3, 2, 5, 0).cpu().numpy()
    assert _close(got37, want37, 1e-5)


@pytest.mark.parametrize("cap", [1, 17, 64, 300])
def test_nms_max_keep_is_prefix_of_full_sweep(dev, cap):
    """post_max_size stops the greedy sweep early (box_torch_ops.py:511-513 truncates afterwards): the capped
    result is the first `cap` survivors of the uncapped one."""
    from detection_3d_amd import box_ops
    b, s = make_boxes(77, 1500)
    order = np.argsort(-s, kind="stable")
    bs = torch.from_numpy(b[order]).to(dev)
    full, nf = box_ops._nms_sorted(bs, 0.4)
    part, npart = box_ops._nms_sorted(bs, 0.4, cap)
    nf, npart = int(nf), int(npart)
    assert npart == min(cap, nf)
    assert torch.equal(part[:npart], full[:npart])


def test_nms_batched_segments_match_single_lists(dev):
    """d3d_rotate_nms_3d_batched: ragged segments (incl. an empty one) addressed through `order`, size clamp in
    the kernel; every segment must give the oracle's survivor list for its own boxes."""
    from detection_3d_amd import box_ops
    n_all, B = 900, 4
    b, s = make_boxes(5, n_all)
    rng = np.random.RandomState(0)
    counts = np.array([700, 0, 333, 64], np.int32)
    order = np.zeros((B, n_all), np.int32)
    segs = []
    for k in range(B):
        sel = rng.permutation(n_all)[:counts[k]]
        sel = sel[np.argsort(-s[sel], kind="stable")]
        order[k, :counts[k]] = sel
        order[k, counts[k]:] = rng.randint(0, n_all, n_all - counts[k])          # garbage past the count
        segs.append(sel)
    aug = (0.25, 0.35)
    keep, nk = box_ops.nms_3d_batched(torch.from_numpy(b).to(dev), torch.from_numpy(order).to(dev),
                                      torch.from_numpy(counts).to(dev), 800, 0.4, aug, 500)
    keep, nk = keep.cpu().numpy(), nk.cpu().numpy()
    bc = b.copy()
    bc[:, 3:5] = np.maximum(bc[:, 3:5], aug[0])
    bc[:, 5] = np.maximum(bc[:, 5], aug[1])
    for k in range(B):
        sel = segs[k]
        want = sel[oracle.rotate_nms_3d(bc[sel], s[sel], 0.4)][:500] if len(sel) else np.zeros(0, np.int64)
        assert nk[k] == len(want)
        assert np.array_equal(keep[k, :nk[k]], want)
    assert nk[0] < counts[0]


def test_roi_prepare_equals_the_host_side_chain_bit_for_bit(dev):
    """d3d_roi_prepare (pixels + RoI format + FPN level in one launch) vs convert_metric_to_pixel ->
    convert_to_roi_format -> Pooler.map_levels, the tensor-op mirror of poolers_3d.py: identical bits, 1 and 3 levels."""
    from detection_3d_amd.detector import Pooler, convert_to_roi_format
    from detection_3d_amd.roi_align_rotated_3d import roi_prepare
    g = torch.Generator().manual_seed(5)
    n = 5000
    b = torch.zeros(n, 7)
    b[:, 0:3] = torch.rand(n, 3, generator=g) * torch.tensor([20.0, 25.0, 3.0])
    b[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 0.6, 2.9]) + 0.001
    b[:, 6] = (torch.rand(n, generator=g) - 0.5) * 7.0                 # beyond [-pi, pi]: limit_period is exercised
    b[:7, 3:5] = torch.tensor([[0.32, 0.1], [0.64, 0.64], [1.28, 0.2], [2.56, 0.1], [0.001, 0.001], [5.12, 5.12], [10.24, 1.0]])
    b = b.to(dev)
    for scales in ([0.25], [0.25, 0.125, 0.0625], [0.5, 0.25]):
        pooler = Pooler((7, 7, 2), scales, 2, 8.0)
        p = b.clone()
        p[:, 0:6] *= 50.0
        want_rois = convert_to_roi_format(p)
        got_rois, got_levels = roi_prepare(b, 50.0, scales, 8.0)
        assert torch.equal(got_rois, want_rois)
        # and against the oracle port (poolers_3d.py:57-69,107-124 restated on the CPU): RoI rows and FPN levels, exact
        from oracle.detector_port import roi_levels, rois_from_boxes
        p_np = b.cpu().numpy().copy()
        p_np[:, 0:6] *= np.float32(50.0)
        assert np.array_equal(got_rois.cpu().numpy(), rois_from_boxes(p_np))
        if len(scales) > 1:
            assert np.array_equal(got_levels.cpu().numpy().astype(np.int64), roi_levels(p_np, scales, 8.0))
        if len(scales) > 1:
            assert torch.equal(got_levels.long(), pooler.map_levels(p))
            assert len(torch.unique(got_levels)) > 1
        else:
            assert got_levels is None
    assert roi_prepare(b[:0], 50.0, [0.25, 0.125], 8.0)[0].shape == (0, 8)


def test_decode_of_selected_rows_and_padded_survivor_list(dev):
    """d3d_box_decode_rows == decode of the gathered rows (oracle, bit-exact; rpn/inference_3d.py:109-123) and
    d3d_gather_kept == boxes[keep[:n]] with clamped sizes, padded with candidate 0, the count stored to a pinned word
    (rpn/inference_3d.py:127-131 + BoxList3D.clamp_size); empty and full survivor lists included."""
    from detection_3d_amd import box_ops
    from detection_3d_amd._lib import host_word
    rng = np.random.RandomState(5)
    n = 5000
    anchors = np.concatenate([rng.rand(n, 3) * 20, 0.2 + rng.rand(n, 3) * 3, (rng.rand(n, 1) - 0.5) * 3], 1).astype(np.float32)
    enc = (rng.randn(n, 7) * 0.3).astype(np.float32)
    rows = rng.permutation(n)[:2000].astype(np.int64)
    got = box_ops.box_decode_rows(torch.from_numpy(enc).to(dev), torch.from_numpy(anchors).to(dev),
                                  torch.from_numpy(rows).to(dev))
    assert np.array_equal(got.cpu().numpy(), oracle.box_decode(enc[rows], anchors[rows]))
    boxes = got.clone()
    boxes[::7, 3] = 0.0002                                       # sizes below the clamp
    boxes[3::11, 5] = -1.0
    scores = torch.from_numpy(rng.rand(2000).astype(np.float32)).to(dev)
    keep = torch.from_numpy(rng.permutation(2000).astype(np.int32)).to(dev)
    word, stored = host_word(dev, "test")
    for nk in (0, 1, 617, 1000):
        cnt = torch.tensor([nk], dtype=torch.int32, device=dev)
        word[0] = -5
        ob, os_ = box_ops.gather_kept(boxes, scores, keep, cnt, 1000, 0.001, count_host=word)
        stored.record()
        stored.synchronize()
        assert int(word[0]) == nk
        want = boxes[keep[:nk].long()].clone()
        want[:, 3:6] = torch.clamp(want[:, 3:6], min=0.001)
        assert torch.equal(ob[:nk], want) and torch.equal(os_[:nk], scores[keep[:nk].long()])
        pad = boxes[0].clone()
        pad[3:6] = torch.clamp(pad[3:6], min=0.001)
        assert bool((ob[nk:] == pad).all()) and bool((os_[nk:] == scores[0]).all())


def _stable_topk(v, k):
    """descending, equal values: lower index first"""
    return np.argsort(-v.astype(np.float64), kind="stable")[:k]


@pytest.mark.parametrize("n,k", [(1, 5), (100, 100), (5000, 2000), (160_000, 2000), (160_000, 4096), (70_001, 1000)])
def test_topk_order_is_the_defined_one(dev, n, k):
    """d3d_topk_segments: exactly the k best in descending order with ties broken by the lower index -- on scores full of
    ties, incl. saturated sigmoids (== 1.0) and a threshold that falls inside a run of equal values."""
    from detection_3d_amd import box_ops
    rng = np.random.RandomState(n + k)
    logits = (rng.randn(n) * 4).astype(np.float32)
    logits[rng.rand(n) < 0.02] = 40.0                      # sigmoid saturates to exactly 1.0
    logits[rng.rand(n) < 0.30] = np.float32(0.37)          # a long run of equal values (the k-th falls into it for large k)
    t = torch.from_numpy(logits).to(dev)
    for sig in (False, True):
        tk = box_ops.topk_segments(t, k, sigmoid=sig)
        vals = torch.sigmoid(t).cpu().numpy() if sig else logits
        want = _stable_topk(vals, k)
        m = int(tk["counts"][0])
        assert m == min(k, n)
        got = tk["idx"][0, :m].cpu().numpy()
        if sig:     # the kernel's own sigmoid defines its order: check against ITS values, and those against torch's
            own = box_ops.topk_segments(t, min(n, 4096), sigmoid=True)
            full = np.empty(n, np.float32)
            if n <= 4096:
                full[own["idx"][0, :n].cpu().numpy()] = own["scores"][0, :n].cpu().numpy()
                assert np.array_equal(full, vals)          # same formula, IEEE division: the same bits as torch.sigmoid
        assert np.array_equal(got, want), (n, k, sig, np.nonzero(got != want)[0][:5])
        assert np.array_equal(tk["scores"][0, :m].cpu().numpy(), vals[want])
    assert (np.sort(vals)[::-1][:k] == 1.0).sum() >= min(k, int((logits == 40.0).sum()))


def test_topk_segments_groups_examples_threshold_and_decode(dev):
    """the RPN layout (columns = class groups, example index per element) with the decode fused in, and the box head's
    layout (row-major segments, score threshold count, mapped indices)"""
    from detection_3d_amd import box_ops
    rng = np.random.RandomState(11)
    n, G, B, k = 30_000, 3, 2, 1000
    obj = (rng.randn(n, G) * 3).astype(np.float32)
    obj[::5] = 25.0
    reg = (rng.randn(n, 7 * G) * 0.2).astype(np.float32)
    anchors = np.concatenate([rng.rand(n, 3) * 20, 0.2 + rng.rand(n, 3) * 3, (rng.rand(n, 1) - 0.5) * 3], 1).astype(np.float32)
    ex = np.sort(rng.randint(0, B, n)).astype(np.int32)
    ex[-7:] = 0                                              # not contiguous per example
    tk = box_ops.topk_segments(torch.from_numpy(obj).to(dev), k, n=n, elem_stride=G, group_stride=1, n_groups=G,
                               example=torch.from_numpy(ex).to(dev), n_examples=B, sigmoid=True,
                               reg=torch.from_numpy(reg).to(dev), anchors=torch.from_numpy(anchors).to(dev))
    sig = torch.sigmoid(torch.from_numpy(obj).to(dev)).cpu().numpy()
    for b in range(B):
        rows = np.nonzero(ex == b)[0]
        for g in range(G):
            s_ = b * G + g
            want = rows[_stable_topk(sig[rows, g], k)]
            assert int(tk["counts"][s_]) == len(want) == k
            assert np.array_equal(tk["idx"][s_].cpu().numpy(), want)
            assert np.array_equal(tk["props"][s_].cpu().numpy(), oracle.box_decode(reg[want, 7 * g:7 * g + 7], anchors[want]))
    # box-head layout: nseg rows of K scores, candidates = scores > 0.05, indices mapped to idx * nc + 1 + g
    K, nc = 1000, 4
    prob = rng.rand(K, nc).astype(np.float32)
    prob[:, 2] = np.where(rng.rand(K) < 0.5, np.float32(0.5), prob[:, 2])       # ties
    prob[:, 3] *= 0.04                                                            # nothing above the threshold
    t = torch.from_numpy(prob).to(dev)
    tk = box_ops.topk_segments(t.view(-1)[1:], K, n=K, elem_stride=nc, group_stride=1, n_groups=nc - 1, min_value=0.05,
                               idx_map=(nc, 1, 1), want_idx64=False, want_idx32=True)
    for j in range(1, nc):
        cnt = int((prob[:, j] > 0.05).sum())
        assert int(tk["counts"][j - 1]) == cnt
        want = _stable_topk(prob[:, j], K)[:cnt]
        assert np.array_equal(tk["idx32"][j - 1, :cnt].cpu().numpy(), want * nc + j)
    assert int(tk["counts"][2]) == 0 and int(tk["counts"][1]) > 400
    # empty input
    e = box_ops.topk_segments(torch.zeros(0, device=dev), 10)
    assert int(e["counts"][0]) == 0


@pytest.mark.parametrize("n,pre,post", [(5000, 2000, 1000), (300, 2000, 100), (2500, 1000, 0)])
def test_rotate_nms_3d_reference_shaped_entry(dev, n, pre, post):
    """d3d_rotate_nms_3d(boxes, scores, pre, post, thr, aug_yx, aug_z): unsorted input, top-k with the defined tie order,
    size clamp for the IoU only, survivor indices into the INPUT -- against the oracle's boxlist_nms_3d (nms_clamped)."""
    from detection_3d_amd import box_ops
    from oracle.detector_port import nms_clamped
    b, s = make_boxes(77 + n, n)
    s[::5] = s[2]                                            # ties
    s[1::9] = 1.0                                            # saturated
    got = box_ops.rotate_nms_3d(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), pre, post or None, 0.5,
                                aug_thickness=(0.3, 0.25)).cpu().numpy()
    want = nms_clamped(b, s, 0.5, (0.3, 0.25), post if post else n, pre_max=pre)
    assert np.array_equal(got, want) and 0 < len(got) <= (post or n)

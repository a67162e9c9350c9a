"""Staged GPU parity of the detector tail (RPN head -> anchors -> top-k -> decode -> NMS -> RoIAlign
-> box head -> per-class NMS) against the CPU port.  A random-init network gives near-tied scores, so
each stage is checked on the GPU's own input to that stage (top-k membership of near-ties is not a
property either implementation defines)."""
import numpy as np
import pytest
import torch

import oracle
from oracle.detector_port import OracleDetector, nms_clamped

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def setup(dev):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(1)
    model = build_detection_model(cfg).to(dev).eval()
    with torch.no_grad():                       # spread the scores so that NMS / thresholds bite
        model.rpn.head.cls_logits.weight.mul_(60)
        model.rpn.head.bbox_pred.weight.mul_(20)
        model.roi_heads.box.predictor.cls_score.weight.mul_(40)
        model.roi_heads.box.predictor.bbox_pred.weight.mul_(100)
    pcl = make_scene(3, 40000)
    coords, feats = voxelize(torch.from_numpy(pcl).to(dev), 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    c_ref, f_ref = oracle.voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    assert np.array_equal(coords.cpu().numpy(), c_ref) and np.array_equal(feats.cpu().numpy(), f_ref)
    result, mid = model([coords, feats], return_intermediates=True)
    return cfg, model, OracleDetector(model.state_dict(), cfg), result, mid


def test_rpn_head_and_anchors(setup):
    cfg, model, orc, result, mid = setup
    feats = [f.features for f in mid["rpn_features"]]
    with torch.no_grad():
        obj, reg = model.rpn.head(feats)
    maps = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy()) for f in mid["rpn_features"]]
    from oracle.detector_port import _lin
    import torch.nn.functional as F
    wo, wr = [], []
    for f, _ in maps:
        t = F.relu(_lin(orc.sd, "rpn.head.conv", torch.from_numpy(f)))
        wo.append(_lin(orc.sd, "rpn.head.cls_logits", t).reshape(-1))
        wr.append(_lin(orc.sd, "rpn.head.bbox_pred", t).reshape(-1, 7))
    assert torch.allclose(obj.reshape(-1).cpu(), torch.cat(wo), rtol=1e-3, atol=1e-4)
    assert torch.allclose(reg.cpu(), torch.cat(wr), rtol=1e-3, atol=1e-4)
    anchors = torch.cat(model.rpn.anchor_generator(mid["rpn_features"]), 0).cpu().numpy()
    assert np.array_equal(anchors, orc.anchors([l for _, l in maps]))
    # the one-launch-per-map form (d3d_anchors) and the batched head are the inference path
    assert np.array_equal(model.rpn.anchor_generator.forward_cat(mid["rpn_features"]).cpu().numpy(), anchors)
    with torch.enable_grad():
        obj_l, reg_l = model.rpn.head(feats)                                     # per-level GEMMs
    assert torch.allclose(obj, obj_l.detach(), rtol=1e-4, atol=1e-5) and torch.allclose(reg, reg_l.detach(), rtol=1e-4, atol=1e-5)
    assert anchors.shape[0] == obj.shape[0] == 4 * sum(m[1].shape[0] for m in maps)


def test_rpn_decode_and_nms_exact(setup, dev):
    cfg, model, orc, result, mid = setup
    from detection_3d_amd import box_ops
    with torch.no_grad():
        obj, reg = model.rpn.head([f.features for f in mid["rpn_features"]])
    anchors = torch.cat(model.rpn.anchor_generator(mid["rpn_features"]), 0)
    scores = obj.reshape(-1).sigmoid()
    sk, idx = scores.topk(min(2000, scores.shape[0]), sorted=True)
    props = box_ops.box_decode(reg[idx], anchors[idx])
    want = oracle.box_decode(reg[idx].cpu().numpy(), anchors[idx].cpu().numpy())
    assert np.array_equal(props.cpu().numpy(), want)
    keep = box_ops.nms_3d_clamped(props, sk, 0.5, [0.3, 0.3], max_proposals=1000, flag="rpn_post").cpu().numpy()
    wkeep = nms_clamped(want, sk.cpu().numpy(), 0.5, [0.3, 0.3], 1000)
    assert np.array_equal(keep, wkeep)
    assert 10 < len(keep) < 2000
    keep2 = box_ops.nms_3d_presorted(props, 0.5, [0.3, 0.3], max_proposals=1000, flag="rpn_post").cpu().numpy()
    assert np.array_equal(keep2, wkeep)
    assert np.array_equal(mid["proposals"].cpu().numpy()[:, :3], want[wkeep][:, :3])


def test_roi_pool_box_head_and_postprocess(setup, dev):
    cfg, model, orc, result, mid = setup
    props = mid["proposals"]
    fe = model.roi_heads.box.feature_extractor
    p = props.clone()
    p[:, 0:6] *= 50
    with torch.no_grad():
        pooled = fe.pooler(mid["roi_features"], p)
    roi_w = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy(), None) for f in mid["roi_features"]]
    want_pooled = orc.pool(roi_w, props.cpu().numpy())
    assert pooled.shape == want_pooled.shape
    assert np.abs(pooled.cpu().numpy() - want_pooled).max() < 1e-5 * max(1, np.abs(want_pooled).max())
    with torch.no_grad():
        x = fe(mid["roi_features"], props)
        logits, reg = model.roi_heads.box.predictor(x)
    wl, wr = orc.box_head(want_pooled)
    assert torch.allclose(logits.cpu(), wl, rtol=2e-3, atol=2e-4)
    assert torch.allclose(reg.cpu(), wr, rtol=2e-3, atol=2e-4)
    # post-processing on identical inputs: survivor sets exact, boxes bit exact
    got = model.roi_heads.box.post_processor(logits, reg, props)
    wb, ws, wlab = orc.post(logits.cpu(), reg.cpu(), props.cpu().numpy())
    assert got["bbox3d"].shape[0] == wb.shape[0] > 0
    assert np.array_equal(got["labels"].cpu().numpy(), wlab)
    assert np.array_equal(got["bbox3d"].cpu().numpy(), wb)
    assert np.allclose(got["scores"].cpu().numpy(), ws, atol=1e-6)
    assert got["bbox3d"].shape[0] <= cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG + 5


def test_buildings_in_flight_match_the_serial_loop_to_the_bit(setup, dev):
    """serving.BuildingPipeline: 2 and 3 buildings in flight on their own streams give the serial loop's detections
    to the bit (per-stream arenas / scratch, no shared state), for buildings of different sizes, and an empty list."""
    from detection_3d_amd.serving import BuildingPipeline
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg, model = setup[0], setup[1]
    clouds = [torch.from_numpy(make_scene(10 + i, n)).to(dev) for i, n in enumerate((40000, 25000, 60000, 30000, 40000))]
    serial = []
    for pcl in clouds:
        c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        serial.append(model([c, f]))
    assert sum(r["bbox3d"].shape[0] for r in serial) > 0
    for n in (2, 3):
        pipe = BuildingPipeline(model, cfg, in_flight=n, device=dev)
        for rep in range(2):
            got = pipe.map(clouds)
            torch.cuda.synchronize()
            assert len(got) == len(serial)
            for g, r in zip(got, serial):
                for k in ("bbox3d", "scores", "labels"):
                    assert g[k].shape == r[k].shape and torch.equal(g[k], r[k]), (n, rep, k)
        assert pipe.map([]) == []


def test_two_lane_pass_matches_the_one_stream_pass_to_the_bit(setup, dev):
    """FPN_Net._forward_two_lane (grid chain on a side stream, per-stream arena lanes) vs the one-stream pass: every
    backbone map and the detections are identical to the bit; a grid requested on the wrong stream is refused."""
    from detection_3d_amd import _lib
    from detection_3d_amd.sparseconvnet import SCN, fpn_net
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg, model = setup[0], setup[1]
    pcl = torch.from_numpy(make_scene(21, 50000)).to(dev)
    outs = {}
    try:
        for two in (False, True, True):
            fpn_net.TWO_LANE = two
            c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
            res, mid = model([c, f], return_intermediates=True)
            torch.cuda.synchronize()
            got = [m.features.clone() for m in mid["rpn_features"] + mid["roi_features"]] + \
                  [res["bbox3d"], res["scores"], res["labels"], mid["proposals"]]
            if two in outs:
                assert all(torch.equal(a, b) for a, b in zip(outs[two], got))
            outs[two] = got
    finally:
        fpn_net.TWO_LANE = True
    assert len(outs[False]) == len(outs[True])
    for a, b in zip(outs[False], outs[True]):
        assert a.shape == b.shape and torch.equal(a, b)
    # the guard: with a geometry stream set, a new grid on another stream is an error, not a race
    c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    net = model.backbone.layers_in[0]([c, f])
    side = torch.cuda.Stream(device=dev)
    net.metadata.set_geometry_stream(side.cuda_stream)
    size = net.spatial_size
    with pytest.raises(_lib.D3DError, match="geometry stream"):
        SCN.Convolution_prepare(size, (size - 2) // 2 + 1, [2, 2, 2], [2, 2, 2], net.metadata)
    with torch.cuda.stream(side):
        assert SCN.Convolution_prepare(size, (size - 2) // 2 + 1, [2, 2, 2], [2, 2, 2], net.metadata) > 0
    net.metadata.set_geometry_stream(None)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_points", [60, 700, 5000])
def test_tiny_buildings_run_through_every_path(setup, dev, n_points):
    """Few points (coarse levels of one or two sites, far fewer anchors than the pre-NMS top-k, possibly no detection):
    the two-lane pass, the one-stream pass and the staged pipeline all run and agree to the bit."""
    from detection_3d_amd.serving import BuildingPipeline
    from detection_3d_amd.sparseconvnet import fpn_net
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg, model = setup[0], setup[1]
    pcl = torch.from_numpy(make_scene(31, 40000)[:n_points].copy()).to(dev)
    res = {}
    try:
        for two in (True, False):
            fpn_net.TWO_LANE = two
            c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
            res[two] = model([c, f])
            torch.cuda.synchronize()
    finally:
        fpn_net.TWO_LANE = True
    piped = BuildingPipeline(model, cfg, in_flight=2, device=dev).map([pcl, pcl, pcl])
    torch.cuda.synchronize()
    for other in [res[False]] + piped:
        for k in ("bbox3d", "scores", "labels"):
            assert other[k].shape == res[True][k].shape and torch.equal(other[k], res[True][k]), (n_points, k)
    assert res[True]["bbox3d"].shape[1] == 7 and torch.isfinite(res[True]["bbox3d"]).all()


def test_post_processing_glue_launches_equal_the_tensor_op_chain(setup, dev):
    """PostProcessor.forward (d3d_post_scores / order / gather around the batched NMS) vs _select_reference, the
    same selection in tensor ops: identical detections, with many and with few survivors (top-k padding case)."""
    cfg, model, _, _, mid = setup
    box = model.roi_heads.box
    x = box.feature_extractor(mid["roi_features"], mid["proposals"])
    logits, reg = box.predictor(x)
    post = box.post_processor
    import torch.nn.functional as F
    from detection_3d_amd import box_ops
    for scale, thresh in ((1.0, post.score_thresh), (0.05, post.score_thresh), (1.0, 0.999)):
        old = post.score_thresh
        post.score_thresh = thresh
        try:
            got = post(logits * scale, reg, mid["proposals"])
            prob = F.softmax(logits * scale, -1)
            want = post._select_reference(prob, box_ops.box_decode(reg, mid["proposals"], post.weights))
        finally:
            post.score_thresh = old
        for k in ("bbox3d", "scores", "labels"):
            assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (scale, thresh, k)
    assert post(logits, reg, mid["proposals"])["bbox3d"].shape[0] > 0


@pytest.mark.parametrize("nseg,n_max,D", [(1, 1, 1), (3, 1000, 200), (4, 700, 100), (2, 50, 200), (8, 1024, 200), (3, 1000, 0),
                                          (5, 333, 1)])
def test_post_select_against_numpy(dev, nseg, n_max, D):
    """d3d_post_select (the cut to detections_per_img + final gathers, box_head_3d/inference.py:140-148) against the same
    selection in numpy: bit-exact rows in the same order, with scores quantised so that ties sit on the threshold (all
    of them stay), segments that kept nothing, fewer survivors than D (the threshold is then a padding entry), D = 0."""
    from detection_3d_amd._lib import check, lib, ptr, stream_of
    rng = np.random.RandomState(nseg * 1000 + n_max + D)
    nbox, nc = nseg * n_max * 2 + 3, nseg + 1
    prob = (np.round(rng.rand(nbox) * 40) / 40 + 0.0125).astype(np.float32)
    boxes = rng.randn(nbox, 7).astype(np.float32)
    nk = rng.randint(0, n_max + 1, nseg).astype(np.int32)
    nk[rng.randint(nseg)] = 0
    if (nseg, n_max) == (2, 50):
        nk[:] = [30, 20]                                   # 50 survivors < D
    keep = rng.randint(0, nbox, (nseg, n_max)).astype(np.int32)
    valid = np.arange(n_max)[None, :] < nk[:, None]
    s = np.where(valid, prob[keep], np.float32(-1)).ravel()
    flat = np.where(valid, keep, 0).ravel()
    thresh = max(np.sort(s)[::-1][D - 1], np.float32(0)) if 0 < D < s.size else np.float32(0)
    sel = s >= thresh
    want_b, want_s, want_l = boxes[flat[sel]], prob[flat[sel]], (flat[sel] % nc).astype(np.int64)
    N = nseg * n_max
    t = lambda a: torch.from_numpy(a).to(dev)
    out_b = torch.full((N, 7), float("nan"), device=dev)
    out_s = torch.full((N,), float("nan"), device=dev)
    out_l = torch.full((N,), -7, dtype=torch.int64, device=dev)
    out_n = torch.full((1,), -1, dtype=torch.int32, device=dev)
    k_, nk_, p_, b_ = t(keep), t(nk), t(prob), t(boxes)
    check(lib().d3d_post_select(ptr(k_), ptr(nk_), nseg, n_max, ptr(p_), ptr(b_), nc, D, ptr(out_b), ptr(out_s), ptr(out_l),
                                ptr(out_n), stream_of()))
    n = int(out_n.item())
    assert n == int(sel.sum())
    assert np.array_equal(out_b[:n].cpu().numpy(), want_b)            # bit-exact
    assert np.array_equal(out_s[:n].cpu().numpy(), want_s)
    assert np.array_equal(out_l[:n].cpu().numpy(), want_l)
    if 0 < D < s.size and int((s >= 0).sum()) >= D:
        assert n >= D                                                 # ties on the threshold all stay


@pytest.mark.parametrize("n_points", [40, 300, 2000])
def test_tiny_scenes_pass_through_the_threaded_schedule(dev, n_points):
    """A handful of points (coarse levels with a single site, rulebooks of one row, offset-split launches of one block)
    through the default pass -- grid chain and rulebook views on the library's threads -- and back to a full-size
    building on the same recycled metadata handles: finite detections, no hang."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(dev).eval()
    with torch.no_grad():
        for n in (n_points, 60000, n_points):
            pcl = torch.from_numpy(make_scene(3, n)).to(dev)
            r = model(list(voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)))
            assert r["bbox3d"].shape[1] == 7 and torch.isfinite(r["bbox3d"]).all() and torch.isfinite(r["scores"]).all()
            assert r["labels"].shape[0] == r["scores"].shape[0] == r["bbox3d"].shape[0] <= cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG * 4


@pytest.mark.parametrize("config,rows", [("4c_Fpn432", [1000, 333, 37]), ("4c_Fpn432", [5]), ("4c_Fpn432", [64, 0, 31]),
                                         ("3G6c_Fpn4321", [700, 129, 64, 1])])
def test_rpn_head_one_launch(dev, config, rows):
    """d3d_rpn_head (the inference path of RPNHead) against the CPU port's per-map linear layers (the reference's
    SingleConvRPNHead_Sparse3D, rpn_sparse3d.py:80-131): ragged maps, an empty map, a single map, one and three
    class groups (32 and 96 output columns); updated weights are re-packed."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import RPNHead
    from oracle.detector_port import _lin
    import torch.nn.functional as F
    cfg = get_cfg(config)
    torch.manual_seed(7)
    head = RPNHead(cfg, cfg.MODEL.BACKBONE.OUT_CHANNELS, 4).to(dev).eval()
    for scale in (1.0, 3.0):
        with torch.no_grad():
            for p in head.parameters():
                p.mul_(scale).add_(0.01 * torch.randn_like(p))
        feats = [torch.randn(n, cfg.MODEL.BACKBONE.OUT_CHANNELS, device=dev) for n in rows]
        assert head._fused_ok(feats)
        obj, reg = head(feats)
        sd = {"h." + k: v for k, v in head.state_dict().items()}
        wo, wr = [], []
        for f in feats:
            t = F.relu(_lin(sd, "h.conv", f.cpu()))
            wo.append(_lin(sd, "h.cls_logits", t).reshape(-1, head.seperate_rpn))
            wr.append(_lin(sd, "h.bbox_pred", t).reshape(-1, 7 * head.seperate_rpn))
        wo, wr = torch.cat(wo), torch.cat(wr)
        assert obj.shape == wo.shape and reg.shape == wr.shape
        # 1e-4 relative to the tensor's magnitude (BASELINE.json), K = 128 fp32 sums in a different order
        assert (obj.cpu() - wo).abs().max() <= 1e-4 * wo.abs().max()
        assert (reg.cpu() - wr).abs().max() <= 1e-4 * wr.abs().max()
        with torch.enable_grad():                                               # the training path: library GEMMs
            obj_l, reg_l = head(feats)
        assert torch.allclose(obj, obj_l.detach(), rtol=1e-4, atol=1e-5 * float(wo.abs().max()))
        assert torch.allclose(reg, reg_l.detach(), rtol=1e-4, atol=1e-5 * float(wr.abs().max()))


@pytest.mark.parametrize("config,rows", [("4c_Fpn432", 1000), ("4c_Fpn432", 37), ("3G6c_Fpn4321", 513), ("4c_Fpn432", 1)])
def test_box_head_mlp_one_launch(dev, config, rows, monkeypatch):
    """d3d_mlp_heads: relu(fc7(relu(fc6 output))) and the predictor's cls_score / bbox_pred
    (roi_box_feature_extractors.py:110-117, roi_box_predictors.py:33-55) in ONE launch, against the CPU port's linear
    layers (1e-4 of the tensor's magnitude: 512-term fp32 sums in another order); the launch with both stages gives the
    bits of fc7 alone followed by the predictor alone (what the modules do when they are called one by one); updated
    weights are re-packed; with gradients enabled the library GEMMs run."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd import detector
    from detection_3d_amd.detector import ROIBoxHead3D
    from oracle.detector_port import _lin
    import torch.nn.functional as F
    monkeypatch.setattr(detector, "_FUSED_BOX_MLP", True)     # (off by default: measured slower than the library GEMMs)
    cfg = get_cfg(config)
    torch.manual_seed(11)
    box = ROIBoxHead3D(cfg).to(dev).eval()
    fe, pred = box.feature_extractor, box.predictor
    assert "_heads" not in dict(fe.named_modules()) and len([k for k in box.state_dict() if "_heads." in k]) == 0
    for scale in (1.0, 2.0):
        with torch.no_grad():
            for p in list(fe.fc7.parameters()) + list(pred.parameters()):
                p.mul_(scale).add_(0.01 * torch.randn_like(p))
            h6 = torch.randn(rows, fe.fc7.in_features, device=dev) * 2
            x = fe._fc7_and_heads(h6)
            assert getattr(x, "_d3d_heads", None) is not None
            logits, reg = pred(x)                                  # the pair that rode on x
            assert logits.data_ptr() == x._d3d_heads[1].data_ptr()
            x2 = x.clone()                                         # (no attribute: the predictor's own launch)
            logits2, reg2 = pred(x2)
            assert torch.equal(logits, logits2) and torch.equal(reg, reg2)
            object.__setattr__(fe, "_heads", None)
            try:
                x3 = fe._fc7_and_heads(h6)                         # fc7 alone
            finally:
                object.__setattr__(fe, "_heads", pred)
            assert torch.equal(x, x3) and getattr(x3, "_d3d_heads", None) is None
        sd = {"f." + k: v for k, v in fe.state_dict().items()}
        sd.update({"p." + k: v for k, v in pred.state_dict().items()})
        wx = F.relu(_lin(sd, "f.fc7", F.relu(h6.cpu())))
        wl, wr = _lin(sd, "p.cls_score", wx), _lin(sd, "p.bbox_pred", wx)
        assert logits.shape == wl.shape and reg.shape == wr.shape
        assert (x.cpu() - wx).abs().max() <= 1e-4 * wx.abs().max()
        assert (logits.cpu() - wl).abs().max() <= 1e-4 * wl.abs().max()
        assert (reg.cpu() - wr).abs().max() <= 1e-4 * wr.abs().max()
        with torch.enable_grad():                                               # the training path: library GEMMs
            xl = F.relu(fe.fc7(F.relu(h6)))
            ll, rl = pred(xl)
        assert getattr(xl, "_d3d_heads", None) is None
        assert torch.allclose(logits, ll.detach(), rtol=1e-4, atol=1e-5 * float(wl.abs().max()))
        assert torch.allclose(reg, rl.detach(), rtol=1e-4, atol=1e-5 * float(wr.abs().max()))


def test_deferred_proposal_count_matches_the_read_back_path(setup, dev, monkeypatch):
    """The inference tail with the RPN's survivor count left on the device while the pooler is enqueued
    (PaddedProposals, d3d_roi_prepare_counted) against the path that reads the count back first: proposals, their
    scores and the detections are identical to the bit; the padding rows are switched off (level -1, zero RoIs)."""
    from detection_3d_amd import detector
    from detection_3d_amd.roi_align_rotated_3d import roi_prepare
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg, model = setup[0], setup[1]
    pcl = torch.from_numpy(make_scene(33, 45000)).to(dev)
    got = {}
    for defer in (True, False):
        monkeypatch.setattr(detector, "_DEFER_PROPOSALS", defer)
        c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        seen = []
        orig = detector.ROIBoxHead3D.forward_padded
        monkeypatch.setattr(detector.ROIBoxHead3D, "forward_padded",
                            lambda self, *a, _o=orig, **k: (seen.append(1), _o(self, *a, **k))[1])
        res, mid = model([c, f], return_intermediates=True)
        monkeypatch.setattr(detector.ROIBoxHead3D, "forward_padded", orig)
        assert bool(seen) == defer                              # the deferred path is the one that ran (or not)
        got[defer] = [res["bbox3d"], res["scores"], res["labels"], mid["proposals"], mid["objectness"]]
    assert got[True][3].shape[0] > 0
    for a, b in zip(got[True], got[False]):
        assert a.shape == b.shape and torch.equal(a, b)
    # d3d_roi_prepare_counted: real rows as d3d_roi_prepare, the others level -1 and zero RoIs
    head = model.roi_heads.box.feature_extractor
    boxes = got[True][3][:50].contiguous()
    pad = torch.cat([boxes, boxes[:14]], 0)
    for n in (50, 17, 0):
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        rois, levels = roi_prepare(pad, head.voxel_scale, head.pooler.scales, head.pooler.canonical_size, None, cnt)
        r0, l0 = roi_prepare(pad, head.voxel_scale, head.pooler.scales, head.pooler.canonical_size)
        assert torch.equal(rois[:n], r0[:n]) and torch.equal(levels[:n], l0[:n])
        assert bool((levels[n:] == -1).all()) and bool((rois[n:] == 0).all())


def test_pooler_one_launch_for_all_levels(setup, dev, monkeypatch):
    """d3d_roi_align_rotated_3d_sparse_forward_levels (every RoI pooled from the map of its level in one launch) against
    one launch per level: the pooled tensor is identical to the bit, padding rows (level -1) stay untouched."""
    from detection_3d_amd import detector
    cfg, model, mid = setup[0], setup[1], setup[4]
    head = model.roi_heads.box.feature_extractor
    props = mid["proposals"]
    assert props.shape[0] > 8 and len(head.pooler.scales) > 1
    cnt = torch.tensor([props.shape[0] - 5], dtype=torch.int32, device=dev)
    got = {}
    for one in (True, False):
        monkeypatch.setattr(detector, "_ROI_ONE_LAUNCH", one)
        full = head.pooler.pool_metric(mid["roi_features"], props, head.voxel_scale, channels_inner=True)
        marked = torch.full_like(full, -7.0)
        orig_empty = torch.empty
        monkeypatch.setattr(torch, "empty", lambda *a, **k: marked if (a and tuple(a[0]) == tuple(full.shape)) else orig_empty(*a, **k))
        part = head.pooler.pool_metric(mid["roi_features"], props, head.voxel_scale, channels_inner=True, count=cnt)
        monkeypatch.setattr(torch, "empty", orig_empty)
        got[one] = (full, part.clone())
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    n = int(cnt.item())
    assert torch.equal(got[True][1][:n], got[True][0][:n]) and bool((got[True][1][n:] == -7.0).all())
    assert float(got[True][0].abs().max()) > 0


def test_anchors_of_all_maps_in_one_launch(setup, monkeypatch):
    """d3d_anchors_maps against one d3d_anchors launch per map and against the oracle's anchors: identical to the bit."""
    from detection_3d_amd import detector
    cfg, model, orc, result, mid = setup
    gen = model.rpn.anchor_generator
    got = {}
    for one in (True, False):
        monkeypatch.setattr(detector, "_ANCHORS_ONE_LAUNCH", one)
        got[one] = gen.forward_cat(mid["rpn_features"]).cpu().numpy()
    want = orc.anchors([f.get_spatial_locations().cpu().numpy() for f in mid["rpn_features"]])
    assert np.array_equal(got[True], got[False]) and np.array_equal(got[True], want)

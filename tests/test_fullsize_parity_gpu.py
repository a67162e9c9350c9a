"""End-to-end parity at BASELINE.json's full size against the CPU oracle port (oracle/detector_port.py does a 500 k-point
building in a few seconds on the box's host cores):

* config #2 (4c_Fpn432 inference, 500 k points): every map the backbone hands to the RPN and to the pooler, compared
  row by row after sorting by coordinate, then the tail stage by stage on the GPU's own input to each stage (a random-init
  network has near-tied scores; top-k membership among ties is not a property either side defines);
* config #3 (6c_Fpn4321 training, 500 k points): one forward + backward + SGD step -- finite losses and gradients, a
  forward that is deterministic to the bit, and dW / dInput of a 256 -> 256 convolution of the coarse levels (the
  k_conv_dw<256,256> and k_conv<128,2,256> instantiations) against oracle.rule_conv_backward on the tensors the step
  itself produced.

Tolerances: integer results (site coordinates per map, NMS survivor lists, labels) exact; fp32 feature maps 2e-4 of the
map's largest magnitude after ~45 chained layers (1e-4 per op is asserted by the per-op tests); decode bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from oracle.detector_port import OracleDetector, nms_clamped
from tests.helpers import nbr_to_rules, sort_by_loc

pytestmark = pytest.mark.gpu
N_POINTS = 500_000


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _sharpen(model):
    with torch.no_grad():                       # spread the scores so that NMS / thresholds bite
        model.rpn.head.cls_logits.weight.mul_(60)
        model.rpn.head.bbox_pred.weight.mul_(20)
        model.roi_heads.box.predictor.cls_score.weight.mul_(40)
        model.roi_heads.box.predictor.bbox_pred.weight.mul_(100)


@pytest.fixture(scope="module")
def full4c(dev):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(2)
    model = build_detection_model(cfg).to(dev).eval()
    _sharpen(model)
    pcl = make_scene(7, N_POINTS)
    with torch.no_grad():
        coords, feats = voxelize(torch.from_numpy(pcl).to(dev), 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        result, mid = model([coords, feats], return_intermediates=True)
    torch.cuda.synchronize()
    c_ref, f_ref = oracle.voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    assert np.array_equal(coords.cpu().numpy(), c_ref) and np.array_equal(feats.cpu().numpy(), f_ref)
    orc = OracleDetector(model.state_dict(), cfg)
    rpn_w, roi_w = orc.fpn(c_ref, f_ref)
    return cfg, model, orc, result, mid, rpn_w, roi_w


def test_backbone_maps_vs_oracle_at_full_size(full4c):
    cfg, model, orc, result, mid, rpn_w, roi_w = full4c
    assert len(mid["rpn_features"]) == len(rpn_w) == 4 and len(mid["roi_features"]) == len(roi_w) == 2
    pairs = list(zip(mid["rpn_features"], [(f, l) for f, l in rpn_w])) + \
        list(zip(mid["roi_features"], [(f, l) for f, l, _ in roi_w]))
    rows = []
    for got, (wf, wl) in pairs:
        gl = got.get_spatial_locations().cpu().numpy()
        gf, gl = sort_by_loc(got.features.cpu().numpy(), gl)
        wf, wl = sort_by_loc(wf, wl)
        assert np.array_equal(gl, wl.astype(np.int64))               # the same set of active sites, exact
        err = rel_err(gf, wf)
        assert err < 2e-4, (gf.shape, err)
        rows.append(gf.shape[0])
    assert max(rows) > 10_000                                        # the 256 x 256 x 32 map of a 500 k-point building


def test_tail_stages_vs_oracle_at_full_size(full4c, dev):
    import torch.nn.functional as F
    from detection_3d_amd import box_ops
    from oracle.detector_port import _lin
    cfg, model, orc, result, mid, rpn_w, roi_w = full4c
    with torch.no_grad():
        # RPN head + anchors on the GPU's maps
        obj, reg = model.rpn.head([f.features for f in mid["rpn_features"]])
        maps = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy()) for f in mid["rpn_features"]]
        wo, wr = [], []
        for f, _ in maps:
            t = F.relu(_lin(orc.sd, "rpn.head.conv", torch.from_numpy(f)))
            wo.append(_lin(orc.sd, "rpn.head.cls_logits", t).reshape(-1))
            wr.append(_lin(orc.sd, "rpn.head.bbox_pred", t).reshape(-1, 7))
        assert torch.allclose(obj.reshape(-1).cpu(), torch.cat(wo), rtol=1e-3, atol=1e-4)
        assert torch.allclose(reg.cpu(), torch.cat(wr), rtol=1e-3, atol=1e-4)
        anchors = model.rpn.anchor_generator.forward_cat(mid["rpn_features"])
        assert np.array_equal(anchors.cpu().numpy(), orc.anchors([l for _, l in maps]))
        assert anchors.shape[0] > 30_000
        # top-k -> decode (bit exact) -> NMS (survivor list exact) on the GPU's scores
        scores = obj.reshape(-1).sigmoid()
        sk, idx = scores.topk(2000, sorted=True)
        props = box_ops.box_decode(reg[idx], anchors[idx])
        want = oracle.box_decode(reg[idx].cpu().numpy(), anchors[idx].cpu().numpy())
        assert np.array_equal(props.cpu().numpy(), want)
        keep = box_ops.nms_3d_presorted(props, 0.5, [0.3, 0.3], max_proposals=1000, flag="rpn_post").cpu().numpy()
        wkeep = nms_clamped(want, sk.cpu().numpy(), 0.5, [0.3, 0.3], 1000)
        assert np.array_equal(keep, wkeep) and 10 < len(keep) <= 1000
        assert np.array_equal(mid["proposals"].cpu().numpy()[:, :3], want[wkeep][:, :3])
        # pooler on the GPU's roi maps and proposals
        props = mid["proposals"]
        fe = model.roi_heads.box.feature_extractor
        p = props.clone()
        p[:, 0:6] *= 50
        pooled = fe.pooler(mid["roi_features"], p)
        roi_g = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy(), None) for f in mid["roi_features"]]
        want_pooled = orc.pool(roi_g, props.cpu().numpy())
        assert np.abs(pooled.cpu().numpy() - want_pooled).max() < 1e-5 * max(1, np.abs(want_pooled).max())
        # box head + post-processing
        x = fe(mid["roi_features"], props)
        logits, regb = model.roi_heads.box.predictor(x)
        wl, wrb = orc.box_head(want_pooled)
        assert torch.allclose(logits.cpu(), wl, rtol=2e-3, atol=2e-4)
        assert torch.allclose(regb.cpu(), wrb, rtol=2e-3, atol=2e-4)
        got = model.roi_heads.box.post_processor(logits, regb, props)
        wb, ws, wlab = orc.post(logits.cpu(), regb.cpu(), props.cpu().numpy())
        assert got["bbox3d"].shape[0] == wb.shape[0] > 0
        assert np.array_equal(got["labels"].cpu().numpy(), wlab)
        assert np.array_equal(got["bbox3d"].cpu().numpy(), wb)
        assert np.allclose(got["scores"].cpu().numpy(), ws, atol=1e-6)
        for k in ("bbox3d", "scores", "labels"):                     # and that is what the detector returned
            assert torch.equal(got[k], result[k])


def test_6c_training_step_at_full_size(dev):
    """configs/6c fpn4321 bs=1 fp32 training, one 500 k-point building (BASELINE.json configs[2])."""
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd import training as T
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene, make_targets
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("6c_Fpn4321")
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(dev).train()
    pcl = torch.from_numpy(make_scene(5, N_POINTS)).to(dev)
    coords, feats = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    boxes, labels = make_targets(5)
    targets = {"bbox3d": torch.from_numpy(boxes).to(dev), "labels": torch.from_numpy(labels).to(dev)}
    opt = T.make_optimizer(cfg, model)

    # the second 256 -> 256 submanifold convolution of scale 6 and the strided 256 -> 256 convolution into scale 7:
    # record what the step feeds them (input rows, gradient of the output rows)
    sub = model.backbone.m_downs[6][1][1][3]
    down = model.backbone.m_downs[7][0][1]
    assert (sub.nIn, sub.nOut, down.nIn, down.nOut) == (256, 256, 256, 256)
    seen = {}

    def capture(name):
        def hook(mod, inp, out):
            x = inp[0]
            seen[name] = {"x": x.features.detach().clone(), "loc": x.get_spatial_locations().cpu().numpy(),
                          "oloc": out.get_spatial_locations().cpu().numpy(), "size": [int(v) for v in x.spatial_size]}
            out.features.register_hook(lambda g: seen[name].__setitem__("g", g.detach().clone()))
            x.features.register_hook(lambda g: seen[name].__setitem__("dx", g.detach().clone()))
        return hook

    h1, h2 = sub.register_forward_hook(capture("sub")), down.register_forward_hook(capture("down"))
    try:
        losses = model([coords, feats], targets)
    finally:
        h1.remove()
        h2.remove()
    assert set(losses) == {"loss_objectness", "loss_rpn_box_reg", "loss_classifier_roi", "loss_box_reg_roi"}
    total = sum(losses.values())
    assert torch.isfinite(total), losses
    opt.zero_grad()
    total.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(g).all() for g in grads.values())
    for k in ("backbone.layers_in.1.weight", "backbone.m_downs.8.1.1.3.weight", "backbone.m_mergeds.3.weight",
              "rpn.head.conv.weight", "roi_heads.box.feature_extractor.fc6.weight"):
        assert grads[k].abs().sum() > 0, k
    assert model.backbone.m_ups[7][1].weight.grad is None            # unconsumed top-down level: no gradient

    # dW / dInput of the two 256 -> 256 convolutions against the oracle, on the tensors of this very step
    s = seen["sub"]
    assert s["x"].shape[0] > 300 and s["x"].shape[1] == 256
    nbr, _ = oracle.subm_nbr(s["loc"].astype(np.int32), [3, 3, 3])
    w = sub.weight.detach().cpu().numpy().reshape(27, 256, 256)
    d_x, d_w = oracle.rule_conv_backward(s["x"].cpu().numpy(), w, nbr_to_rules(nbr), s["g"].cpu().numpy())
    assert rel_err(sub.weight.grad.cpu().numpy().reshape(27, 256, 256), d_w) < 2e-4
    # the input of `sub` is a BatchNorm output that nothing else consumes: its gradient is this convolution's dInput
    assert rel_err(s["dx"].cpu().numpy(), d_x) < 2e-4
    d = seen["down"]
    lo, ru = oracle.conv_rules(d["loc"].astype(np.int32), [2, 2, 2], [2, 2, 2], [v // 2 for v in d["size"]])
    assert np.array_equal(lo.astype(np.int64), d["oloc"])            # first-touch numbering of the coarse grid
    wd = down.weight.detach().cpu().numpy().reshape(8, 256, 256)
    d_x2, d_wd = oracle.rule_conv_backward(d["x"].cpu().numpy(), wd, ru, d["g"].cpu().numpy())
    assert rel_err(down.weight.grad.cpu().numpy().reshape(8, 256, 256), d_wd) < 2e-4
    assert rel_err(d["dx"].cpu().numpy(), d_x2) < 2e-4

    before = model.backbone.m_downs[3][1][1][1].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, model.backbone.m_downs[3][1][1][1].weight)
    torch.cuda.synchronize()

    # the forward of the training graph is deterministic to the bit (no float atomics, fixed reduction orders)
    with torch.no_grad():
        a = model.backbone([coords, feats])
        b = model.backbone([coords, feats])
    for ta, tb in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(ta.features, tb.features)
    assert a[0][0].features.shape[1] == 128 and len(a[0]) == 6 and len(a[1]) == 2
    del scn


def test_config5_bf16_batch_of_four_1m_point_buildings(dev):
    """BASELINE configs[4] at its stated size: bf16 sparse conv, ~1 M-point scenes (35 x 27 x 2.7 m), 4 examples per batch.
    Against the batch-aware oracle port at full size: site coordinates of every map handed on exact, the fp32 pass 2e-4 and
    the bf16 pass 5e-2 of each map's magnitude (the tolerance of tests/test_bf16_gpu.py for a chain of layers);
    deterministic to the bit; every example yields detections."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(4)
    model = build_detection_model(cfg).to(dev).eval()
    _sharpen(model)
    B, cs, fs, cref = 4, [], [], []
    for b in range(B):
        pcl = make_scene(900 + b, 1_000_000, (35.0, 27.0, 2.7))
        with torch.no_grad():
            c, f = voxelize(torch.from_numpy(pcl).to(dev), 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        cs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64, device=dev)], 1))
        fs.append(f)
        cr, _ = oracle.voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        cref.append(np.concatenate([cr, np.full((cr.shape[0], 1), b, np.int64)], 1))
    coords, feats = torch.cat(cs), torch.cat(fs)
    assert coords.shape[0] > 3_500_000 and np.array_equal(coords.cpu().numpy(), np.concatenate(cref))
    with torch.no_grad():
        res32, mid32 = model([coords, feats, B], return_intermediates=True)
        model.backbone.compute_dtype = torch.bfloat16
        try:
            res16, mid16 = model([coords, feats, B], return_intermediates=True)
            res16b, mid16b = model([coords, feats, B], return_intermediates=True)
        finally:
            model.backbone.compute_dtype = torch.float32
    torch.cuda.synchronize()
    # site coordinates of the maps: exact against the oracle's rule builders at full size
    _, loc = oracle.input_sites(np.concatenate(cref))
    size = list(cfg.SPARSE3D.VOXEL_FULL_SCALE)
    locs = [loc]
    for _k in range(5):
        size = [v // 2 for v in size]
        lo, _ = oracle.conv_rules(locs[-1], [2, 2, 2], [2, 2, 2], size)
        locs.append(lo)
    roi16 = mid16["roi_features"]                       # ups[4] = scale 4 (256 x 256 x 32), ups[3] = scale 5
    for t, want in zip(roi16, (locs[4], locs[5])):
        assert np.array_equal(t.get_spatial_locations().cpu().numpy(), want.astype(np.int64))
    # every map handed on, against the batch-aware oracle port at full size (one backbone pass over the 4 examples, BatchNorm
    # over all rows as the reference; ~30 s on the host): the fp32 pass at 2e-4 of the map's magnitude, the bf16 pass at the
    # chain tolerance of tests/test_bf16_gpu.py (5e-2: one rounding to bf16 per layer, ~45 layers)
    orc = OracleDetector(model.state_dict(), cfg)
    rpn_w, roi_w = orc.fpn(np.concatenate(cref), feats.cpu().numpy())
    want_maps = [(f, l) for f, l in rpn_w] + [(f, l) for f, l, _ in roi_w]
    got16, got32 = mid16["rpn_features"] + roi16, mid32["rpn_features"] + mid32["roi_features"]
    assert len(want_maps) == len(got16) == len(got32) == 6
    for a, c, (wf, wl) in zip(got16, got32, want_maps):
        wf, wl = sort_by_loc(wf, wl)
        g32, l32 = sort_by_loc(c.features.cpu().numpy(), c.get_spatial_locations().cpu().numpy())
        g16, l16 = sort_by_loc(a.features.cpu().numpy(), a.get_spatial_locations().cpu().numpy())
        assert np.array_equal(l32, wl.astype(np.int64)) and np.array_equal(l16, l32)      # site sets incl. example index
        assert rel_err(g32, wf) < 2e-4, (g32.shape, rel_err(g32, wf))
        assert rel_err(g16, wf) < 5e-2, (g16.shape, rel_err(g16, wf))
    for a, b, c in zip(got16, mid16b["rpn_features"] + mid16b["roi_features"], got32):
        assert torch.equal(a.features, b.features)                                   # deterministic to the bit
        assert a.features.dtype == torch.float32 and a.features.shape == c.features.shape
        assert rel_err(a.features.cpu().numpy(), c.features.cpu().numpy()) < 5e-2
    assert len(res16) == B and all(r["bbox3d"].shape[0] > 0 for r in res16) and all(r["bbox3d"].shape[0] > 0 for r in res32)
    for r, rb in zip(res16, res16b):
        assert torch.equal(r["bbox3d"], rb["bbox3d"]) and torch.equal(r["scores"], rb["scores"])


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: 3G6c_Fpn4321 (three class groups) at 500 k points against the oracle port's grouped detector
def _targets_3g(dev):
    from detection_3d_amd.synthetic import make_targets
    b, l = make_targets(5)
    b, l = b.copy(), l.copy()
    extra = np.array([[12.5, 9.5, 0.0, 19.0, 25.0, 0.1, 0.0], [12.5, 9.5, 2.6, 19.0, 25.0, 0.1, 0.0],      # floor, ceiling
                      [6.0, 5.0, 0.0, 8.0, 6.0, 0.1, 0.0], [18.0, 13.0, 2.6, 8.0, 6.0, 0.1, 0.0]], np.float32)
    b = np.concatenate([b, extra])
    l = np.concatenate([l, np.array([5, 4, 5, 4], np.int64)])                  # suncg_metas order: ceiling 4, floor 5
    return b, l, {"bbox3d": torch.from_numpy(b).to(dev), "labels": torch.from_numpy(l).to(dev)}


@pytest.fixture(scope="module")
def full3g(dev):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg("3G6c_Fpn4321")
    torch.manual_seed(3)
    model = build_detection_model(cfg).to(dev).eval()
    _sharpen(model)
    pcl = make_scene(5, N_POINTS)
    with torch.no_grad():
        coords, feats = voxelize(torch.from_numpy(pcl).to(dev), 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        result, mid = model([coords, feats], return_intermediates=True)
    torch.cuda.synchronize()
    c_ref, f_ref = oracle.voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    assert np.array_equal(coords.cpu().numpy(), c_ref) and np.array_equal(feats.cpu().numpy(), f_ref)
    orc = OracleDetector(model.state_dict(), cfg)
    assert orc.groups == [[0, 2, 3], [6, 1], [7, 4, 5]]
    rpn_w, roi_w = orc.fpn(c_ref, f_ref)
    return cfg, model, orc, result, mid, rpn_w, roi_w, (coords, feats)


def test_3g6c_backbone_maps_vs_oracle_at_full_size(full3g):
    cfg, model, orc, result, mid, rpn_w, roi_w, _ = full3g
    assert len(mid["rpn_features"]) == len(rpn_w) == 6 and len(mid["roi_features"]) == len(roi_w) == 2
    pairs = list(zip(mid["rpn_features"], [(f, l) for f, l in rpn_w])) + \
        list(zip(mid["roi_features"], [(f, l) for f, l, _ in roi_w]))
    for got, (wf, wl) in pairs:
        gf, gl = sort_by_loc(got.features.cpu().numpy(), got.get_spatial_locations().cpu().numpy())
        wf, wl = sort_by_loc(wf, wl)
        assert np.array_equal(gl, wl.astype(np.int64))
        assert rel_err(gf, wf) < 2e-4, gf.shape


def test_3g6c_grouped_tail_vs_oracle_at_full_size(full3g, dev):
    """Per class group: proposals (top-k -> decode -> NMS survivor list) exact against the port's selector on the device's
    scores; pooler / box head within tolerance; final boxes and original-class labels of the three groups bit-exact."""
    import torch.nn.functional as F
    cfg, model, orc, result, mid, rpn_w, roi_w, _ = full3g
    G = 3
    pre, post = cfg.MODEL.RPN.FPN_PRE_NMS_TOP_N_TEST, cfg.MODEL.RPN.FPN_POST_NMS_TOP_N_TEST
    assert (pre, post, cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG) == (1000, 500, 100)   # x 0.5 per group
    with torch.no_grad():
        obj, reg = model.rpn.head([f.features for f in mid["rpn_features"]])
        assert obj.shape[1] == G and reg.shape[1] == 7 * G
        maps = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy()) for f in mid["rpn_features"]]
        wo, wr, wanchors = orc.rpn_head(maps)
        assert torch.allclose(obj.cpu(), wo, rtol=1e-3, atol=1e-4) and torch.allclose(reg.cpu(), wr, rtol=1e-3, atol=1e-4)
        anchors = model.rpn.anchor_generator.forward_cat(mid["rpn_features"])
        assert np.array_equal(anchors.cpu().numpy(), wanchors) and anchors.shape[0] > 40_000
        props, sep_id = mid["proposals"], mid["sep_id"]
        assert sep_id is not None and props.shape[0] == sep_id.shape[0]
        scores = obj.sigmoid().cpu()
        n_seen = 0
        for g in range(G):
            want_p, want_s, keep, _ = orc.rpn_select(None, reg[:, 7 * g:7 * g + 7].cpu(), wanchors, scores=scores[:, g])
            want_p = want_p.copy()
            want_p[:, 3:6] = np.maximum(want_p[:, 3:6], 0.001)                       # BoxList3D.clamp_size
            got_p = props[sep_id == g].cpu().numpy()
            assert 10 < len(keep) <= post
            assert np.array_equal(got_p, want_p), g                                   # survivor list and decode, exact
            n_seen += len(keep)
        assert n_seen == props.shape[0]
        # pooler on the device's maps and proposals, box head, grouped post-processing
        fe = model.roi_heads.box.feature_extractor
        p = props.clone()
        p[:, 0:6] *= 50
        pooled = fe.pooler(mid["roi_features"], p)
        roi_g = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy(), None) for f in mid["roi_features"]]
        want_pooled = orc.pool(roi_g, props.cpu().numpy())
        assert np.abs(pooled.cpu().numpy() - want_pooled).max() < 1e-5 * max(1, np.abs(want_pooled).max())
        x = fe(mid["roi_features"], props)
        logits, regb = model.roi_heads.box.predictor(x)
        assert logits.shape[1] == 8 and regb.shape[1] == 56
        wl, wrb = orc.box_head(want_pooled)
        assert torch.allclose(logits.cpu(), wl, rtol=2e-3, atol=2e-4) and torch.allclose(regb.cpu(), wrb, rtol=2e-3, atol=2e-4)
        # (selection on the device's own probabilities: softmax over each group's columns, rows of that group)
        probs = [F.softmax(logits[sep_id == g][:, torch.tensor(cols, device=dev)], -1).cpu().numpy()
                 for g, cols in enumerate(orc.groups)]
        wb, ws, wlab = orc.post_grouped(logits.cpu(), regb.cpu(), props.cpu().numpy(), sep_id.cpu().numpy(), probs)
        got = model.roi_heads.box(mid["roi_features"], props, sep_id=sep_id)
        assert got["bbox3d"].shape[0] == wb.shape[0] > 0
        assert np.array_equal(got["labels"].cpu().numpy(), wlab) and set(wlab.tolist()).issubset({1, 2, 3, 4, 5})
        gb = got["bbox3d"].cpu().numpy()
        bad = np.nonzero((gb != wb).any(1))[0]
        assert len(bad) == 0, (len(bad), bad[:8], gb[bad[:3]], wb[bad[:3]], got["scores"].cpu().numpy()[bad[:3]], ws[bad[:3]])
        assert np.array_equal(got["scores"].cpu().numpy(), ws)
        for k in ("bbox3d", "scores", "labels"):
            assert torch.equal(got[k], result[k])


def test_3g6c_training_step_and_group_labels_at_full_size(full3g, dev):
    """One 3G6c training step on the 500 k-point building: the three groups' target sets and RPN label sets
    (rpn/loss_3d.py:88-109 per group, seperate_classifier.py:83-95) against the port's Matcher on the oracle IoU."""
    from detection_3d_amd import box_ops
    from detection_3d_amd import training as T
    from oracle import detector_port as P
    from tests.helpers import label_differences_sit_on_thresholds
    cfg, model, orc, result, mid, rpn_w, roi_w, (coords, feats) = full3g
    b, l, targets = _targets_3g(dev)
    sep = model.rpn.sep
    tg = sep.group_targets(targets)
    want_tg = P.group_targets(orc.groups, b, l)
    with torch.no_grad():
        anchors = model.rpn.anchor_generator.forward_cat(mid["rpn_features"])
    a_np = anchors.cpu().numpy()
    lossf = model.rpn.loss_evaluator
    rpn = cfg.MODEL.RPN
    n_diff_total = 0
    for g in range(3):
        assert np.array_equal(tg[g]["bbox3d"].cpu().numpy(), want_tg[g][0])
        assert np.array_equal(tg[g]["labels"].cpu().numpy(), want_tg[g][1])
        gt = tg[g]["bbox3d"]
        labels, _ = lossf.prepare_targets(anchors, gt)
        labels = labels.cpu().numpy()
        want, (q_o, yaw_o, _) = P.rpn_labels(cfg, a_np, want_tg[g][0], return_iou=True)
        # (a) the device's Matcher on the device's own IoU = the port's Matcher on the same matrix, exactly
        q_g = box_ops.boxes_iou_3d(gt, anchors, lossf.aug, criterion=2, flag='rpn_label_generation')
        yaw_g = torch.abs(box_ops.limit_period(gt[:, -1].view(-1, 1) - anchors[:, -1].view(1, -1), 0.5, np.pi))
        m = P.matcher(q_g.cpu().numpy(), rpn.FG_IOU_THRESHOLD, rpn.BG_IOU_THRESHOLD, True, yaw_g.cpu().numpy(), rpn.YAW_THRESHOLD)
        same = (m >= 0).astype(np.float32)
        same[m == -2] = -1
        assert np.array_equal(labels, same), g
        # (b) against the oracle IoU: the matrices agree to 1e-4, and a label differs only where an IoU sits on a threshold
        assert np.array_equal(yaw_g.cpu().numpy(), yaw_o)
        assert np.abs(q_g.cpu().numpy() - q_o).max() <= 1e-4
        mask_g = (yaw_g.cpu().numpy() < np.float32(rpn.YAW_THRESHOLD)).astype(np.float32)
        n_diff, bad = label_differences_sit_on_thresholds(labels, want, q_g.cpu().numpy() * mask_g, q_o * mask_g,
                                                          rpn.BG_IOU_THRESHOLD, rpn.FG_IOU_THRESHOLD)
        assert not bad, (g, n_diff, bad[:5])
        n_diff_total += n_diff
        assert (labels == 1).sum() >= max(1, gt.shape[0] // 2), g
    assert n_diff_total <= 1e-3 * 3 * a_np.shape[0]
    # the step itself
    model.train()
    try:
        opt = T.make_optimizer(cfg, model)
        losses = model([coords, feats], targets)
        assert len(losses) == 12 and all(torch.isfinite(v) for v in losses.values()), losses
        opt.zero_grad()
        sum(losses.values()).backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        assert model.rpn.head.cls_logits.weight.grad.abs().sum() > 0
        assert model.roi_heads.box.predictor.cls_score.weight.grad.abs().sum() > 0
    finally:
        model.eval()

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
    Integer results against the oracle at full size (site coordinates of every map handed on: exact); bf16 maps against the
    fp32 pass of the same batch (5e-2 of the map's magnitude, the tolerance of tests/test_bf16_gpu.py for a chain of
    layers); deterministic to the bit; every example yields detections."""
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
    for a, b, c in zip(mid16["rpn_features"] + roi16, mid16b["rpn_features"] + mid16b["roi_features"],
                       mid32["rpn_features"] + mid32["roi_features"]):
        assert torch.equal(a.features, b.features)                                   # deterministic to the bit
        assert a.features.dtype == torch.float32 and a.features.shape == c.features.shape
        assert rel_err(a.features.cpu().numpy(), c.features.cpu().numpy()) < 5e-2
    assert len(res16) == B and all(r["bbox3d"].shape[0] > 0 for r in res16) and all(r["bbox3d"].shape[0] > 0 for r in res32)
    for r, rb in zip(res16, res16b):
        assert torch.equal(r["bbox3d"], rb["bbox3d"]) and torch.equal(r["scores"], rb["scores"])

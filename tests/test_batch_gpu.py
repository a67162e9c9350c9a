"""Several examples per batch (BASELINE.json configs[4]: bs=4) and several class groups per example (configs[3]: 3G6c)
through the detector tail as SEGMENTS of one launch set.

The reference loops: examples in rpn/inference_3d.py:92-163 and box_head_3d/inference.py:66-99, class groups in
modeling/seperate_classifier.py:58-95,299-321.  Here every (example, group) is a segment of one top-k / decode / batched NMS /
post-processing launch set.  Checked: (1) against the batch-aware CPU oracle port stage by stage on the GPU's own input to
each stage, (2) against this package's single-example / single-group path called once per segment -- to the bit (the
backbone's BatchNorm and the box head's BatchNorm3d see the whole batch in both the reference and here, so a batch is NOT
the concatenation of single-example passes before those stages; after them it is)."""
import numpy as np
import pytest
import torch

import oracle
from oracle.detector_port import OracleDetector
from tests.helpers import sort_by_loc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _sharpen(model):
    model.rpn.head.cls_logits.weight.mul_(60)
    model.rpn.head.bbox_pred.weight.mul_(20)
    model.roi_heads.box.predictor.cls_score.weight.mul_(40)
    model.roi_heads.box.predictor.bbox_pred.weight.mul_(100)


def _batch(dev, cfg, sizes, seed0=40):
    """examples voxelised one by one (each shifted by its own minimum, as the dataset does per scene) and listed one after
    the other with their index in the 4th coordinate column"""
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    cs, fs, cs_ref, fs_ref = [], [], [], []
    for b, n in enumerate(sizes):
        pcl = make_scene(seed0 + b, n)
        c, f = voxelize(torch.from_numpy(pcl).to(dev), 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        cs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64, device=dev)], 1))
        fs.append(f)
        cr, fr = oracle.voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        cs_ref.append(np.concatenate([cr, np.full((cr.shape[0], 1), b, np.int64)], 1))
        fs_ref.append(fr)
    return torch.cat(cs), torch.cat(fs), np.concatenate(cs_ref), np.concatenate(fs_ref)


@pytest.fixture(scope="module")
def batch3(dev):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(3)
    model = build_detection_model(cfg).to(dev).eval()
    with torch.no_grad():
        _sharpen(model)
        coords, feats, c_ref, f_ref = _batch(dev, cfg, (30000, 20000, 26000))
        assert np.array_equal(coords.cpu().numpy(), c_ref) and np.array_equal(feats.cpu().numpy(), f_ref)
        results, mid = model([coords, feats], return_intermediates=True)
    return cfg, model, results, mid, c_ref, f_ref


def test_batch_backbone_and_tail_vs_oracle(batch3, dev):
    cfg, model, results, mid, c_ref, f_ref = batch3
    B = 3
    assert isinstance(results, list) and len(results) == B
    orc = OracleDetector(model.state_dict(), cfg)
    rpn_w, roi_w = orc.fpn(c_ref, f_ref)
    for got, (wf, wl) in list(zip(mid["rpn_features"], [(f, l) for f, l in rpn_w])) + \
            list(zip(mid["roi_features"], [(f, l) for f, l, _ in roi_w])):
        gl = got.get_spatial_locations().cpu().numpy()
        assert set(np.unique(gl[:, 3]).tolist()) == {0, 1, 2}
        assert (np.diff(gl[:, 3]) >= 0).all()                      # sites are numbered example by example
        gf, gl = sort_by_loc(got.features.cpu().numpy(), gl)
        wf, wl = sort_by_loc(wf, wl)
        assert np.array_equal(gl, wl.astype(np.int64))
        assert rel_err(gf, wf) < 2e-4
    # RPN selection per example on the GPU's maps: survivors exact
    props, ex = mid["proposals"], mid["example_id"]
    assert (np.diff(ex.cpu().numpy()) >= 0).all() and set(ex.unique().tolist()) == {0, 1, 2}
    maps = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy()) for f in mid["rpn_features"]]
    for b in range(B):
        maps_b = [(f[loc[:, 3] == b], loc[loc[:, 3] == b]) for f, loc in maps]
        want, _ = orc.rpn(maps_b)
        got = props[ex == b].cpu().numpy()
        assert got.shape == want.shape and got.shape[0] > 10
        assert np.allclose(got[:, :3], want[:, :3], atol=1e-5) and np.allclose(np.maximum(want[:, 3:6], 0.001), got[:, 3:6], atol=1e-5)
    # pooling with the example index in the RoI, box head over all RoIs, post-processing per example
    fe = model.roi_heads.box.feature_extractor
    ids = ex.to(torch.int32).contiguous()
    pooled = fe.pooler.pool_metric(mid["roi_features"], props, 50, channels_inner=False, batch_ids=ids)
    roi_g = [(f.features.cpu().numpy(), f.get_spatial_locations().cpu().numpy(), None) for f in mid["roi_features"]]
    want_pooled = orc.pool(roi_g, props.cpu().numpy(), ex.cpu().numpy(), B)
    assert np.abs(pooled.cpu().numpy() - want_pooled).max() < 1e-5 * max(1, np.abs(want_pooled).max())
    x = fe(mid["roi_features"], props, ids)
    logits, reg = model.roi_heads.box.predictor(x)
    wl_, wr_ = orc.box_head(want_pooled)
    assert torch.allclose(logits.cpu(), wl_, rtol=2e-3, atol=2e-4) and torch.allclose(reg.cpu(), wr_, rtol=2e-3, atol=2e-4)
    for b in range(B):
        m = ex == b
        wb, ws, wlab = orc.post(logits[m].cpu(), reg[m].cpu(), props[m].cpu().numpy())
        r = results[b]
        assert r["bbox3d"].shape[0] == wb.shape[0] > 0
        assert np.array_equal(r["labels"].cpu().numpy(), wlab) and np.array_equal(r["bbox3d"].cpu().numpy(), wb)
        assert np.allclose(r["scores"].cpu().numpy(), ws, atol=1e-6)


def test_batch_segments_equal_the_single_example_calls(batch3, dev):
    """RPN selection and post-processing of the batch == the single-example functions called on each example's rows."""
    cfg, model, results, mid, _, _ = batch3
    rpn, box = model.rpn, model.roi_heads.box
    feats = mid["rpn_features"]
    obj, reg = rpn.head([f.features for f in feats])
    anchors = rpn.anchor_generator.forward_cat(feats)
    A = rpn.anchor_generator.num_anchors_per_location()
    example = torch.cat([f.get_spatial_locations()[:, 3].repeat_interleave(A) for f in feats])
    segs = rpn.select_proposals_segments(obj, reg, anchors, example, 3, False)
    for b in range(3):
        m = example == b
        p1, s1 = rpn.select_proposals(obj[m], reg[m].contiguous(), anchors[m], False)
        assert torch.equal(segs[b][0], p1) and torch.equal(segs[b][1], s1)
    props, ex = mid["proposals"], mid["example_id"]
    x = box.feature_extractor(mid["roi_features"], props, ex.to(torch.int32).contiguous())
    logits, regb = box.predictor(x)
    for b in range(3):
        m = ex == b
        one = box.post_processor(logits[m], regb[m].contiguous(), props[m])
        for k in ("bbox3d", "scores", "labels"):
            assert torch.equal(one[k], results[b][k]), (b, k)


def test_single_example_with_batch_column_equals_plain_input(dev):
    """coords [N,4] with batch index 0 everywhere is the one-example case: same detections as coords [N,3]."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(3)
    model = build_detection_model(cfg).to(dev).eval()
    _sharpen(model)
    coords, feats, _, _ = _batch(dev, cfg, (30000,))
    a = model([coords, feats])
    b = model([coords[:, :3].contiguous(), feats])
    assert isinstance(a, dict)
    for k in ("bbox3d", "scores", "labels"):
        assert torch.equal(a[k], b[k])


def test_3g6c_group_segments_equal_the_group_loop(dev):
    """3G6c inference: the three class groups as segments of one launch set == the reference's per-group loop spelled
    with this package's single-group functions (seperate_classifier.py:58-95,299-321), to the bit."""
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    cfg = get_cfg("3G6c_Fpn4321")
    torch.manual_seed(5)
    model = build_detection_model(cfg).to(dev).eval()
    _sharpen(model)
    coords, feats, _, _ = _batch(dev, cfg, (50000,), seed0=60)
    res, mid = model([coords[:, :3].contiguous(), feats], return_intermediates=True)
    rpn, box, sep = model.rpn, model.roi_heads.box, model.rpn.sep
    assert sep.group_num == 3
    obj, reg = rpn.head([f.features for f in mid["rpn_features"]])
    anchors = rpn.anchor_generator.forward_cat(mid["rpn_features"])
    props, ids = [], []
    for gi in range(3):
        p, _ = rpn.select_proposals(obj[:, gi], reg[:, 7 * gi:7 * gi + 7].contiguous(), anchors, False)
        p = p.clone()
        p[:, 3:6] = torch.clamp(p[:, 3:6], min=0.001)
        props.append(p)
        ids.append(torch.full((p.shape[0],), gi, dtype=torch.int64, device=dev))
    props, ids = torch.cat(props), torch.cat(ids)
    assert torch.equal(props, mid["proposals"])
    x = box.feature_extractor(mid["roi_features"], props)
    logits, regb = box.predictor(x)
    ids_g = [torch.nonzero(ids == gi).view(-1) for gi in range(3)]
    parts = []
    for gi, (lg, rg) in enumerate(zip(sep.seperate_pred_logits(logits, ids_g), sep.seperate_pred_box(regb, ids_g))):
        r = box.post_processor(lg, rg.contiguous(), props[ids_g[gi]])
        r["labels"] = sep.org_label(gi, r["labels"])
        parts.append(r)
    want = {k: torch.cat([r[k] for r in parts]) for k in ("bbox3d", "scores", "labels")}
    assert want["bbox3d"].shape[0] > 0 and set(want["labels"].unique().tolist()).issubset({1, 2, 3, 4, 5})
    for k in ("bbox3d", "scores", "labels"):
        assert torch.equal(res[k], want[k]), k

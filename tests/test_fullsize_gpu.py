"""Parity at BASELINE.json's full size (500 k-point building, full scale 4096 x 4096 x 512, config 4c): direct
comparison with the oracle where the oracle finishes in seconds (a1-a5, first convolution), and size-independent
properties for the rest of the path (determinism to the bit, linearity, NMS idempotence, encode/decode round trip)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from tests.helpers import canon_rules, nbr_to_rules

pytestmark = pytest.mark.gpu
SIZE = (4096, 4096, 512)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def scene(dev):
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    pcl = make_scene(5, 500_000)
    coords, feats = voxelize(torch.from_numpy(pcl).to(dev), 50, SIZE)
    return pcl, coords, feats


def test_voxel_sites_and_means_exact_at_full_size(scene, dev):
    from detection_3d_amd import sparseconvnet as scn
    pcl, coords, feats = scene
    c_ref, f_ref = oracle.voxelize(pcl, 50, SIZE)
    assert coords.shape[0] == c_ref.shape[0] > 450_000
    assert np.array_equal(coords.cpu().numpy(), c_ref) and np.array_equal(feats.cpu().numpy(), f_ref)
    t = scn.InputLayer(3, SIZE, mode=4)([coords, feats])
    sop, loc = oracle.input_sites(c_ref)
    assert np.array_equal(t.get_spatial_locations().cpu().numpy(), loc.astype(np.int64))   # first-occurrence ids
    assert np.array_equal(t.features.cpu().numpy(), oracle.input_forward(f_ref, sop, loc.shape[0], True))


def test_rulebooks_exact_at_full_size(scene, dev):
    from detection_3d_amd import sparseconvnet as scn
    from detection_3d_amd._lib import check, ints, lib, stream_of
    pcl, coords, feats = scene
    t = scn.InputLayer(3, SIZE, mode=4)([coords, feats])
    m = t.metadata
    _, loc = oracle.input_sites(coords.cpu().numpy())
    nr = ctypes.c_long(0)
    check(lib().d3d_subm_prepare(m._h, ints(SIZE), ints([3, 3, 3]), stream_of(), ctypes.byref(nr)))
    nbr, total = oracle.subm_nbr(loc, [3, 3, 3])
    assert nr.value == total > 900_000
    assert np.array_equal(canon_rules(m.export_rules(0, SIZE, [3, 3, 3]).cpu().numpy()), canon_rules(nbr_to_rules(nbr)))
    out_size = [s // 2 for s in SIZE]
    n_out, nr = ctypes.c_int(0), ctypes.c_long(0)
    check(lib().d3d_conv_prepare(m._h, ints(SIZE), ints(out_size), ints([2, 2, 2]), ints([2, 2, 2]), stream_of(),
                                 ctypes.byref(n_out), ctypes.byref(nr)))
    lo, ru = oracle.conv_rules(loc, [2, 2, 2], [2, 2, 2], out_size)
    assert n_out.value == lo.shape[0] and nr.value == ru.shape[0] == loc.shape[0]
    assert np.array_equal(m.getSpatialLocations(out_size).cpu().numpy(), lo.astype(np.int64))
    assert np.array_equal(canon_rules(m.export_rules(1, SIZE, [2, 2, 2], [2, 2, 2]).cpu().numpy()), canon_rules(ru))


def test_first_convolution_vs_oracle_and_linearity_at_full_size(scene, dev):
    from detection_3d_amd import sparseconvnet as scn
    pcl, coords, feats = scene
    t = scn.InputLayer(3, SIZE, mode=4)([coords, feats])
    torch.manual_seed(3)
    conv = scn.SubmanifoldConvolution(3, 9, 32, 3, False).to(dev)
    y = conv(t).features
    _, loc = oracle.input_sites(coords.cpu().numpy())
    nbr, _ = oracle.subm_nbr(loc, [3, 3, 3])
    w = conv.weight.detach().cpu().numpy()
    want = oracle.nbr_conv(t.features.cpu().numpy(), w.reshape(w.shape[0], w.shape[2], w.shape[3]), nbr)
    assert rel_err(y.cpu().numpy(), want) < 1e-4
    # linearity on a wide layer: conv(a x + b z) == a conv(x) + b conv(z)
    conv2 = scn.SubmanifoldConvolution(3, 32, 32, 3, False).to(dev)
    x = scn.SparseConvNetTensor(y, t.metadata, t.spatial_size)
    z = scn.SparseConvNetTensor(torch.randn_like(y), t.metadata, t.spatial_size)
    mix = scn.SparseConvNetTensor(0.75 * x.features - 1.5 * z.features, t.metadata, t.spatial_size)
    lhs = conv2(mix).features
    rhs = 0.75 * conv2(x).features - 1.5 * conv2(z).features
    assert float((lhs - rhs).abs().max()) < 1e-4 * float(rhs.abs().max())


def test_detector_is_deterministic_to_the_bit_at_full_size(scene, dev):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    pcl, coords, feats = scene
    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(dev).eval()
    with torch.no_grad():
        model.rpn.head.cls_logits.weight.mul_(60)
        model.rpn.head.bbox_pred.weight.mul_(20)
        model.roi_heads.box.predictor.cls_score.weight.mul_(40)
    r1, mid1 = model([coords, feats], return_intermediates=True)
    r2, mid2 = model([coords, feats], return_intermediates=True)
    for a, b in zip(mid1["rpn_features"] + mid1["roi_features"], mid2["rpn_features"] + mid2["roi_features"]):
        assert torch.equal(a.features, b.features)                 # no float atomics, fixed reduction orders
    assert torch.equal(mid1["proposals"], mid2["proposals"])
    for k in ("bbox3d", "scores", "labels"):
        assert torch.equal(r1[k], r2[k])
    assert r1["bbox3d"].shape[0] > 0
    # the same building through the one-stream pass and through two buildings in flight: identical to the bit
    from detection_3d_amd.serving import BuildingPipeline
    from detection_3d_amd.sparseconvnet import fpn_net
    try:
        fpn_net.TWO_LANE = False
        r3, mid3 = model([coords, feats], return_intermediates=True)
    finally:
        fpn_net.TWO_LANE = True
    for a, b in zip(mid1["rpn_features"] + mid1["roi_features"], mid3["rpn_features"] + mid3["roi_features"]):
        assert torch.equal(a.features, b.features)
    # every schedule of the pass -- grid chain on library threads with / without the rulebook views on a third stream
    # (the default is the first), the chain driven by the launch thread on two streams, that with the views on a third --
    # runs the same kernels on the same data: identical maps and detections
    saved = (fpn_net.ASYNC_GEOMETRY, fpn_net.ASYNC_VIEWS, fpn_net.PLAN_LANE)
    try:
        for mode in ((True, False, False), (False, False, False), (False, False, True)):
            fpn_net.ASYNC_GEOMETRY, fpn_net.ASYNC_VIEWS, fpn_net.PLAN_LANE = mode
            r4, mid4 = model([coords, feats], return_intermediates=True)
            for a, b in zip(mid1["rpn_features"] + mid1["roi_features"], mid4["rpn_features"] + mid4["roi_features"]):
                assert torch.equal(a.features, b.features), mode
            for k in ("bbox3d", "scores", "labels"):
                assert torch.equal(r1[k], r4[k]), mode
    finally:
        fpn_net.ASYNC_GEOMETRY, fpn_net.ASYNC_VIEWS, fpn_net.PLAN_LANE = saved
    cloud = torch.from_numpy(pcl).to(dev)
    piped = BuildingPipeline(model, cfg, in_flight=2, device=dev).map([cloud, cloud, cloud])
    torch.cuda.synchronize()
    for other in [r3] + piped:
        for k in ("bbox3d", "scores", "labels"):
            assert torch.equal(r1[k], other[k])
    # NMS idempotence on the detector's own proposals (2000 candidates): survivors survive again, all of them
    from detection_3d_amd import box_ops
    props = mid1["proposals"]
    keep = box_ops.nms_3d_presorted(props, cfg.MODEL.RPN.NMS_THRESH, cfg.MODEL.RPN.NMS_AUG_THICKNESS_Y_Z,
                                    max_proposals=1000, flag="rpn_post")
    assert keep.numel() == props.shape[0] and torch.equal(keep, torch.arange(props.shape[0], device=dev))


def test_box_coder_round_trip_at_anchor_count(scene, dev):
    from detection_3d_amd import box_ops
    from detection_3d_amd.training import box_encode
    g = torch.Generator(device="cpu").manual_seed(1)
    n = 38_000                                                     # anchors of the 4c configuration on this scene
    anchors = torch.rand(n, 7, generator=g)
    anchors[:, 0:3] *= 20
    anchors[:, 3:6] = 0.2 + anchors[:, 3:6] * 3
    anchors[:, 6] = (anchors[:, 6] - 0.5) * 3.0
    boxes = anchors + (torch.rand(n, 7, generator=g) - 0.5) * 0.4
    boxes[:, 3:6] = boxes[:, 3:6].clamp(min=0.05)
    a, b = anchors.to(dev), boxes.to(dev)
    back = box_ops.box_decode(box_encode(b, a), a)
    d = (back - b).abs()
    d[:, 6] = torch.minimum(d[:, 6], (d[:, 6] - np.pi).abs())     # yaw is decoded modulo pi (limit_period)
    assert float(d.max()) < 2e-4

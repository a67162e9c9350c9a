"""Training path on the GPU: one building through forward + backward + SGD; losses finite, every consumed
parameter receives a gradient, the unused top-down levels do not, and a few steps on one scene reduce the loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, name, n_points=60000):
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import make_scene, make_targets
    from detection_3d_amd.voxelize import voxelize
    cfg = get_cfg(name)
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(dev).train()
    pcl = torch.from_numpy(make_scene(5, n_points)).to(dev)
    coords, feats = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    boxes, labels = make_targets(5)
    targets = {"bbox3d": torch.from_numpy(boxes).to(dev), "labels": torch.from_numpy(labels).to(dev)}
    return cfg, model, coords, feats, targets


def test_label_generation_against_oracle_iou(dev):
    import oracle
    from detection_3d_amd import training as T
    cfg, model, coords, feats, targets = _setup(dev, "4c_Fpn432", 30000)
    with torch.no_grad():
        rpn_feats, _ = model.backbone([coords, feats])
        anchors = torch.cat(model.rpn.anchor_generator(rpn_feats), 0)
    lossf = model.rpn.loss_evaluator
    labels, reg = lossf.prepare_targets(anchors, targets["bbox3d"])
    q = oracle.boxes_iou_3d(targets["bbox3d"].cpu().numpy(), anchors.cpu().numpy(), lossf.aug, criterion=2)
    yaw = torch.abs(T.box_ops.limit_period(targets["bbox3d"][:, -1].view(-1, 1) - anchors[:, -1].view(1, -1), 0.5, np.pi))
    want = lossf.matcher(torch.from_numpy(q).to(dev), yaw_diff=yaw)
    want_labels = (want >= 0).float()
    want_labels[want == -2] = -1
    # labels are integer results: a label may differ from the oracle-IoU label only where an IoU (equal on both sides to
    # 1e-4, last-ulp libm cases) sits on one of the Matcher's thresholds
    from tests.helpers import label_differences_sit_on_thresholds
    q_g = T.box_ops.boxes_iou_3d(targets["bbox3d"], anchors, lossf.aug, criterion=2, flag='rpn_label_generation').cpu().numpy()
    assert np.abs(q_g - q).max() <= 1e-4
    mask = (yaw.cpu().numpy() < np.float32(lossf.matcher.yaw_threshold)).astype(np.float32)
    n_diff, bad = label_differences_sit_on_thresholds(labels.cpu().numpy(), want_labels.cpu().numpy(), q_g * mask, q * mask,
                                                      lossf.matcher.low_threshold, lossf.matcher.high_threshold)
    assert not bad and n_diff <= 1e-3 * labels.numel(), (n_diff, bad[:5])
    assert (labels == 1).sum().item() >= targets["bbox3d"].shape[0] // 2


def test_3g6c_grouped_train_and_eval(dev):
    """configs/3G6c: three class groups, each with its own RPN column, proposal budget, background column."""
    from detection_3d_amd import training as T
    from detection_3d_amd.synthetic import make_targets
    cfg, model, coords, feats, targets = _setup(dev, "3G6c_Fpn4321")
    b, l = make_targets(5)
    l = l.copy()
    l[-4:] = [4, 5, 4, 5]                                        # some floor / ceiling boxes
    b[-4:, 3:6] = [[6.0, 8.0, 0.1]] * 4
    targets = {"bbox3d": torch.from_numpy(b).to(dev), "labels": torch.from_numpy(l).to(dev)}
    sep = model.rpn.sep
    assert sep.grouped_classes == [[0, 2, 3], [6, 1], [7, 4, 5]] and sep.total_classes == 8
    tg = sep.group_targets(targets)
    assert [int(t["labels"].max()) for t in tg] == [2, 1, 2]
    opt = T.make_optimizer(cfg, model)
    losses = model([coords, feats], targets)
    assert len(losses) == 12 and all(torch.isfinite(v) for v in losses.values()), losses
    opt.zero_grad()
    sum(losses.values()).backward()
    assert model.rpn.head.cls_logits.weight.grad.abs().sum() > 0
    opt.step()
    model.eval()
    res = model([coords, feats])
    assert res["bbox3d"].shape[0] <= 3 * cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG + 3
    assert set(res["labels"].unique().tolist()).issubset({1, 2, 3, 4, 5})


@pytest.mark.parametrize("name", ["4c_Fpn432", "6c_Fpn4321"])
def test_train_steps(dev, name):
    from detection_3d_amd import training as T
    cfg, model, coords, feats, targets = _setup(dev, name)
    opt = T.make_optimizer(cfg, model)
    torch.manual_seed(1)
    totals = []
    for it in range(4):
        losses = model([coords, feats], targets)
        assert set(losses) == {"loss_objectness", "loss_rpn_box_reg", "loss_classifier_roi", "loss_box_reg_roi"}
        total = sum(losses.values())
        assert torch.isfinite(total), losses
        opt.zero_grad()
        total.backward()
        if it == 0:
            got = {k for k, p in model.named_parameters() if p.grad is not None and p.grad.abs().sum() > 0}
            none = {k for k, p in model.named_parameters() if p.grad is None}
            assert "backbone.layers_in.1.weight" in got and "backbone.m_downs.8.1.1.3.weight" in got
            assert "rpn.head.conv.weight" in got and "roi_heads.box.feature_extractor.fc6.weight" in got
            n_up = max(cfg.MODEL.RPN.RPN_SCALES_FROM_TOP + list(cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_FROM_TOP))
            assert f"backbone.m_mergeds.{n_up - 1}.weight" in got
            if n_up < 8:   # computed-but-unconsumed levels of the reference: no gradient (DDP find_unused_parameters)
                assert f"backbone.m_mergeds.{n_up}.weight" in none and "backbone.m_ups.7.1.weight" in none
            assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        opt.step()
        totals.append(total.item())
    assert totals[-1] < totals[0], totals

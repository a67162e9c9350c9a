"""Config surface: derived fields of intact_cfg (tools/train_net_sparse3d.py:231-323) for the three
target configs (values of SURVEY.md appendix B) and checkpoint key names."""
import os

import pytest
import torch

from detection_3d_amd.config import class_to_label, get_cfg


def test_derived_fields_4c():
    c = get_cfg("4c_Fpn432")
    assert c.MODEL.RPN.RPN_MAP_SIZES == [[256, 256, 32], [128, 128, 16], [64, 64, 8]]
    assert c.MODEL.RPN.ANCHOR_STRIDE == [[32] * 3, [16] * 3, [32] * 3, [64] * 3]
    assert c.MODEL.ROI_BOX_HEAD.POOLER_SCALES_SPATIAL == [1 / 16, 1 / 32]
    assert c.SPARSE3D.SCENE_SIZE == [81.92, 81.92, 10.24]
    assert class_to_label(c.INPUT.CLASSES) == {"background": 0, "wall": 1, "window": 2, "door": 3}
    assert c.SOLVER.TRACK_RUNNING_STATS is False and c.SOLVER.BN_MOMENTUM == 0.95


def test_derived_fields_6c_and_3g6c():
    c = get_cfg("6c_Fpn4321")
    assert c.MODEL.RPN.RPN_MAP_SIZES[-1] == [32, 32, 4]
    assert c.MODEL.RPN.ANCHOR_STRIDE == [[32] * 3, [64] * 3, [128] * 3, [16] * 3, [32] * 3, [64] * 3]
    assert class_to_label(c.INPUT.CLASSES)["floor"] == 4 and class_to_label(c.INPUT.CLASSES)["ceiling"] == 5
    g = get_cfg("3G6c_Fpn4321")
    assert g.MODEL.SEPARATE_CLASSES_ID == [[1], [5, 4]]
    assert (g.MODEL.RPN.FPN_PRE_NMS_TOP_N_TEST, g.MODEL.RPN.FPN_POST_NMS_TOP_N_TEST) == (1000, 500)
    assert (g.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, g.MODEL.ROI_HEADS.DETECTIONS_PER_IMG) == (256, 100)


def test_opts_override_and_unknown_key():
    c = get_cfg("4c_Fpn432", ["MODEL.RPN.NMS_THRESH", "0.6", "SOLVER.IMS_PER_BATCH", 2])
    assert c.MODEL.RPN.NMS_THRESH == 0.6 and c.SOLVER.IMS_PER_BATCH == 2
    with pytest.raises(KeyError):
        get_cfg("4c_Fpn432", ["MODEL.NOPE", "1"])


@pytest.mark.skipif(not os.path.exists("/root/reference/configs"), reason="reference not mounted")
def test_reference_yaml_files_load_unchanged():
    pairs = [("configs/4c/4c_Fpn432_bs1_lr5_SD.yaml", "4c_Fpn432"), ("configs/6c/6c_Fpn4321_bs1_lr5.yaml", "6c_Fpn4321"),
             ("configs/3G6c/3G6c_Fpn4321_bs1_lr2.yaml", "3G6c_Fpn4321")]
    for path, name in pairs:
        a, b = get_cfg(os.path.join("/root/reference", path)), get_cfg(name)
        for sec in ("MODEL", "SPARSE3D", "TEST"):
            assert a[sec] == b[sec], (path, sec)


def test_checkpoint_key_names_and_shapes():
    from detection_3d_amd.detector import build_detection_model
    sd = build_detection_model(get_cfg("4c_Fpn432")).state_dict()
    bb = {k: v for k, v in sd.items() if k.startswith("backbone.")}
    assert len(bb) == 197 and sum(v.numel() for k, v in bb.items() if "running" not in k) == 21147124
    assert tuple(sd["backbone.layers_in.1.weight"].shape) == (27, 1, 9, 32)
    assert tuple(sd["backbone.m_downs.1.0.1.weight"].shape) == (8, 1, 32, 64)
    assert tuple(sd["backbone.m_downs.0.0.1.3.weight"].shape) == (27, 1, 32, 32)
    assert tuple(sd["backbone.m_shortcuts.8.weight"].shape) == (1, 1, 256, 128)
    assert tuple(sd["backbone.m_ups.0.1.weight"].shape) == (8, 1, 128, 128)
    assert tuple(sd["backbone.m_mergeds.7.weight"].shape) == (27, 1, 128, 128)
    assert tuple(sd["backbone.convs_pro2d.0.weight"].shape) == (32, 1, 128, 128)
    assert tuple(sd["rpn.head.conv.weight"].shape) == (128, 128, 1, 1)
    assert tuple(sd["rpn.head.bbox_pred.weight"].shape) == (28, 128, 1, 1)
    assert tuple(sd["roi_heads.box.feature_extractor.conv3d.0.weight"].shape) == (512, 128, 1, 1, 4)
    assert tuple(sd["roi_heads.box.feature_extractor.fc6.weight"].shape) == (512, 24576)
    assert tuple(sd["roi_heads.box.predictor.bbox_pred.weight"].shape) == (28, 512)
    n_params = sum(p.numel() for p in build_detection_model(get_cfg("4c_Fpn432")).parameters())
    assert abs(n_params - 34.3e6) < 0.1e6
    # a checkpoint written with the reference's DDP prefix loads after stripping "module."
    ck = {"module." + k: v for k, v in sd.items()}
    m = build_detection_model(get_cfg("4c_Fpn432"))
    missing = m.load_state_dict({k[len("module."):]: v for k, v in ck.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_seperate_classifier_groups():
    from detection_3d_amd.detector import SeperateClassifier
    g = get_cfg("3G6c_Fpn4321")
    sep = SeperateClassifier(g.MODEL.SEPARATE_CLASSES_ID, len(g.INPUT.CLASSES))
    assert sep.grouped_classes == [[0, 2, 3], [6, 1], [7, 4, 5]]          # seperate_classifier.py:20-36
    assert sep.total_classes == 8 and sep.class_nums == [3, 2, 3]
    t = {"bbox3d": torch.arange(42, dtype=torch.float32).view(6, 7), "labels": torch.tensor([1, 2, 5, 3, 4, 1])}
    tg = sep.group_targets(t)
    assert tg[0]["labels"].tolist() == [1, 2] and tg[1]["labels"].tolist() == [1, 1] and tg[2]["labels"].tolist() == [1, 2]
    assert sep.org_label(2, torch.tensor([1, 2])).tolist() == [4, 5]
    from detection_3d_amd.detector import build_detection_model
    m = build_detection_model(g)
    assert tuple(m.state_dict()["rpn.head.cls_logits.weight"].shape) == (12, 128, 1, 1)
    assert tuple(m.state_dict()["roi_heads.box.predictor.bbox_pred.weight"].shape) == (56, 512)

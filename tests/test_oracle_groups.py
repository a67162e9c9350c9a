"""CPU tests of the oracle port's class-group pieces (3G6c), pinned by arrays the reference's own modules produced in the
build container (tests/golden/make_golden.py -> ref_python.npz / ref_python2.npz): grouping, regrouped targets, per-group
column slices, label map back (modeling/seperate_classifier.py) and the numpy Matcher (modeling/matcher.py)."""
import math
import os

import numpy as np
import torch

from oracle import detector_port as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_port_matcher_matches_reference_outputs():
    g = np.load(os.path.join(GOLD, "ref_python.npz"))
    assert np.array_equal(P.matcher(g["match_iou"], 0.55, 0.2, True, None, math.pi), g["match_res"])
    got = P.matcher(g["match2_iou"], 0.55, 0.2, True, g["match2_yaw"], 0.7)
    assert np.array_equal(got, g["match2_rpn"])
    assert np.array_equal(P.matcher(g["match2_iou"], 0.5, 0.5, False), g["match2_roi"])


def test_port_class_groups_match_reference_fixtures():
    g = np.load(os.path.join(GOLD, "ref_python2.npz"))
    for tag, n_in in (("3g6c", 6), ("sew4c", 4)):
        enc = g[f"sep_{tag}_ids"].tolist()               # make_golden.py: flat class ids, -1, then the group lengths
        cut = enc.index(-1)
        flat_ids, lens = enc[:cut], enc[cut + 1:]
        sep_ids, o = [], 0
        for n in lens:
            sep_ids.append(flat_ids[o:o + n])
            o += n
        nums = g[f"sep_{tag}_class_nums"].tolist()
        flat = g[f"sep_{tag}_grouped_flat"].tolist()
        want, o = [], 0
        for n in nums:
            want.append(flat[o:o + n])
            o += n
        groups = P.class_groups(sep_ids, n_in)
        assert groups == want
        assert int(g[f"sep_{tag}_total"][0]) == n_in + len(groups) - 1
        o2s = g[f"sep_{tag}_org_to_sep"]
        for gi, cols in enumerate(groups):
            for i, c in enumerate(cols):
                assert o2s[c].tolist() == [gi, i]
            assert np.array_equal(np.asarray(cols), g[f"sep_{tag}_back_g{gi}"])
        tg = P.group_targets(groups, g[f"sep_{tag}_boxes_in"], g[f"sep_{tag}_labels_in"])
        for gi, (b, l) in enumerate(tg):
            assert np.array_equal(b, g[f"sep_{tag}_tg{gi}_boxes"]) and np.array_equal(l, g[f"sep_{tag}_tg{gi}_labels"])
        # the per-group slices post_grouped takes (seperate_pred_logits / seperate_pred_box)
        logits, reg, sep_id = (torch.from_numpy(g[f"sep_{tag}_{k}"]) for k in ("logits", "reg", "sep_id"))
        n = logits.shape[0]
        for gi, cols in enumerate(groups):
            idx = torch.nonzero(sep_id == gi).view(-1)
            c = torch.tensor(cols)
            assert np.array_equal(logits[idx][:, c].numpy(), g[f"sep_{tag}_logits_g{gi}"])
            assert np.array_equal(reg.view(n, -1, 7)[:, c, :].reshape(n, -1)[idx].numpy(), g[f"sep_{tag}_reg_g{gi}"])


def test_port_post_grouped_is_the_per_group_post_with_labels_mapped_back():
    from detection_3d_amd.config import get_cfg
    cfg = get_cfg("3G6c_Fpn4321")
    orc = P.OracleDetector({}, cfg)
    assert orc.groups == [[0, 2, 3], [6, 1], [7, 4, 5]] and orc.rpn_groups == 3
    rng = np.random.RandomState(3)
    K = 90
    props = np.concatenate([rng.rand(K, 3) * [20, 15, 0.2], 0.3 + rng.rand(K, 3) * [0.2, 3, 2], (rng.rand(K, 1) - 0.5) * 3], 1).astype(np.float32)
    sep_id = np.repeat(np.arange(3), K // 3)
    logits = torch.from_numpy(rng.randn(K, 8).astype(np.float32) * 3)
    reg = torch.from_numpy(rng.randn(K, 56).astype(np.float32) * 0.1)
    b, s, l = orc.post_grouped(logits, reg, props, sep_id)
    assert b.shape[0] == s.shape[0] == l.shape[0] > 0
    o = 0
    for gi, cols in enumerate(orc.groups):                      # group after group, each the plain post() of its slice
        ids = np.nonzero(sep_id == gi)[0]
        c = torch.tensor(cols)
        bg, sg, lg = orc.post(logits[ids][:, c], reg.view(K, -1, 7)[:, c, :].reshape(K, -1)[ids], props[ids])
        assert np.array_equal(b[o:o + len(sg)], bg) and np.array_equal(s[o:o + len(sg)], sg)
        assert np.array_equal(l[o:o + len(sg)], np.asarray(cols)[lg]) and (lg >= 1).all()
        o += len(sg)
    assert o == len(s) and set(l.tolist()).issubset({1, 2, 3, 4, 5})

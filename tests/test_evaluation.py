"""Detection metric: known answers on the CPU (IoU from the oracle) and HIP-IoU agreement on the GPU."""
import numpy as np
import pytest

import oracle
from detection_3d_amd.config import get_cfg
from detection_3d_amd.evaluation import calc_ap, eval_detection_suncg
from detection_3d_amd.synthetic import make_targets


def _oracle_iou(gt, pred, aug):
    return oracle.boxes_iou_3d(gt, pred, aug, -1)


def _scene(seed, drop=0, jitter=0.0, extra=0):
    rng = np.random.RandomState(seed)
    b, l = make_targets(seed)
    keep = np.arange(len(b))[drop:]
    pb = b[keep] + rng.randn(len(keep), 7).astype(np.float32) * jitter * np.array([1, 1, 1, 0, 0, 0, 0], np.float32)
    pl = l[keep]
    ps = rng.uniform(0.6, 1.0, len(keep)).astype(np.float32)
    if extra:                                   # false positives far away, low scores
        fb = b[:extra].copy()
        fb[:, :2] += 100
        pb, pl, ps = np.concatenate([pb, fb]), np.concatenate([pl, l[:extra]]), np.concatenate([ps, np.full(extra, 0.1, np.float32)])
    return {"bbox3d": pb, "labels": pl, "scores": ps}, {"bbox3d": b, "labels": l}


def test_perfect_and_degraded_predictions():
    cfg = get_cfg("4c_Fpn432")
    preds, gts = zip(*[_scene(s) for s in range(3)])
    r = eval_detection_suncg(preds, gts, cfg, iou_fn=_oracle_iou)
    assert abs(r["map"] - 1.0) < 1e-9 and np.allclose(r["ap"][1:], 1.0)
    # false positives scored below every true positive do not change AP; missing half of the walls does
    preds, gts = zip(*[_scene(s, extra=4) for s in range(3)])
    assert abs(eval_detection_suncg(preds, gts, cfg, iou_fn=_oracle_iou)["map"] - 1.0) < 1e-9
    preds, gts = zip(*[_scene(s, drop=5) for s in range(3)])
    r = eval_detection_suncg(preds, gts, cfg, iou_fn=_oracle_iou)
    assert r["ap"][1] < 0.7 and abs(r["ap"][2] - 1.0) < 1e-9          # 5 of 10 walls missing: recall 0.5 -> AP07 = 6/11
    assert abs(r["ap"][1] - 6 / 11) < 1e-9


def test_voc07_and_area_ap_formulas():
    prec = [None, np.array([1.0, 0.5, 2 / 3, 0.5])]
    rec = [None, np.array([0.5, 0.5, 1.0, 1.0])]
    ap07 = calc_ap(prec, rec, True)[1]
    assert abs(ap07 - (6 * 1.0 + 5 * 2 / 3) / 11) < 1e-9
    assert abs(calc_ap(prec, rec, False)[1] - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-9
    # the reference's recall-precision-score-IoU table (suncg_eval.py:915-942): hand-evaluated rows
    scores = [None, np.array([0.9, 0.8, 0.7, 0.6])]
    pious = [None, np.array([0.8, 0.1, 0.6, 0.2])]
    ap, tab = calc_ap(prec, rec, True, scores, pious)
    assert abs(ap[1] - ap07) < 1e-12 and tab.shape == (2, 11, 4)
    assert np.allclose(tab[1, 0], [0.0, 1.0, 0.91, 0.8])            # recall 0: no rec <= 0 -> max score + 0.01
    assert np.allclose(tab[1, 5], [0.5, 1.0, 0.8, 0.8])             # recall 0.5: lowest score with rec <= 0.5
    assert np.allclose(tab[1, 6], [0.6, 2 / 3, 0.8, 0.6])           # only the last two predictions reach 0.6
    assert np.allclose(tab[1, 10], [1.0, 2 / 3, 0.6, 0.6])
    assert np.allclose(tab[0], tab[1])                               # slot 0 = class mean


@pytest.mark.gpu
def test_metric_with_hip_iou_matches_oracle_iou(dev):
    cfg = get_cfg("6c_Fpn4321")
    preds, gts = zip(*[_scene(s, drop=s % 3, jitter=0.05, extra=2) for s in range(4)])
    a = eval_detection_suncg(preds, gts, cfg)
    b = eval_detection_suncg(preds, gts, cfg, iou_fn=_oracle_iou)
    assert np.allclose(a["ap"], b["ap"], equal_nan=True) and 0.3 < a["map"] <= 1.0


def test_aiou_rows_known_answers():
    """AIoU as the reference reports it (suncg_eval.py:217,306: IoU column of the 11-step table, averaged) and
    cal_mious (:968-981): exact predictions give 1, a known shift gives the analytic IoU of the shifted boxes."""
    from detection_3d_amd.evaluation import cal_mious
    cfg = get_cfg("4c_Fpn432")
    preds, gts = zip(*[_scene(s) for s in range(3)])
    r = eval_detection_suncg(preds, gts, cfg, iou_fn=_oracle_iou)
    assert np.allclose(r["aiou"][1:], 1.0, atol=1e-5) and abs(r["aiou"][0] - 1.0) < 1e-5
    assert np.allclose(r["mious"][1:], 1.0, atol=1e-5) and np.isnan(r["mious"][0])
    # every prediction shifted 2 cm along z: the BEV IoU stays 1, the z IoU of a box of height h is (h - d)/(h + d)
    shifted = []
    for p in preds:
        q = {k: v.copy() for k, v in p.items()}
        q["bbox3d"][:, 2] += 0.02
        shifted.append(q)
    r2 = eval_detection_suncg(shifted, gts, cfg, iou_fn=_oracle_iou)
    want = {}
    for g in gts:
        for l in np.unique(g["labels"]):
            h = np.maximum(g["bbox3d"][g["labels"] == l][:, 5], cfg.TEST.EVAL_AUG_THICKNESS_Z_TAR_ANC[0])
            want.setdefault(int(l), []).append(np.mean((h - 0.02) / (h + 0.02)))
    for l, vals in want.items():
        assert abs(r2["mious"][l] - np.mean(vals)) < 2e-3, (l, r2["mious"][l], np.mean(vals))
        assert 0.9 < r2["aiou"][l] <= 1.0
    assert cal_mious({1: [np.array([0.9, 0.2, 0.7]), np.array([0.5])]}, 0.3, 3)[1] == np.mean([0.8, 0.5])

"""Detection metric of the reference: per-class IoU matching with thickness augmentation,
PASCAL-VOC precision / recall and the VOC-07 11-point AP the README numbers are quoted on.

    eval_detection_suncg          data3d/evaluation/suncg/suncg_eval.py:714-740
    calc_prec_rec                 :750-892   (a prediction matches the gt of maximum IoU; the highest-scored
                                              prediction of a gt is the true positive)
    calc_ap                       :894-966   (use_07_metric=True at suncg_eval.py:85)
The IoU matrices [gt x pred] come from the HIP kernel (box_ops.boxes_iou_3d, flag 'eval'); the bookkeeping is
O(#boxes) numpy on the host, as in the reference.
"""
from collections import defaultdict

import numpy as np
import torch

from . import box_ops


def _gpu_iou(gt, pred, aug):
    dev = torch.device("cuda")
    return box_ops.boxes_iou_3d(torch.as_tensor(gt, device=dev), torch.as_tensor(pred, device=dev), aug, -1,
                                flag='eval').cpu().numpy()


def calc_prec_rec(preds, gts, iou_thresh, eval_aug_thickness, iou_fn=_gpu_iou):
    """preds / gts: lists (one per building) of dicts with numpy 'bbox3d' [n,7] yx_zb, 'labels' [n] (+ 'scores')."""
    n_pos, score, match, predious = defaultdict(int), defaultdict(list), defaultdict(list), defaultdict(list)
    per_building = defaultdict(list)        # label -> [best IoU of every prediction of building i] (cal_mious input)
    for gt, pred in zip(gts, preds):
        pl, ps, pb = np.asarray(pred["labels"]), np.asarray(pred["scores"]), np.asarray(pred["bbox3d"], np.float32)
        gl, gb = np.asarray(gt["labels"]), np.asarray(gt["bbox3d"], np.float32)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            pm = pl == l
            order = ps[pm].argsort()[::-1]
            pb_l, ps_l = pb[pm][order], ps[pm][order]
            gb_l = gb[gl == l]
            n_pos[l] += gb_l.shape[0]
            score[l].extend(ps_l)
            if len(pb_l) == 0:
                continue
            if len(gb_l) == 0:
                match[l].extend((0,) * pb_l.shape[0])
                predious[l].extend((0,) * pb_l.shape[0])
                continue
            iou = iou_fn(gb_l.copy(), pb_l.copy(), eval_aug_thickness)      # [n_gt, n_pred]
            gt_index = iou.argmax(axis=0)
            gt_index[iou.max(axis=0) < iou_thresh] = -1
            predious[l].extend(iou.max(0))
            per_building[l].append(np.asarray(iou.max(0), dtype=np.float64))
            selec = np.zeros(gb_l.shape[0], dtype=bool)
            for gi in gt_index:                                              # predictions in score order
                if gi >= 0:
                    match[l].append(0 if selec[gi] else 1)
                    selec[gi] = True
                else:
                    match[l].append(0)
    n_cls = max(n_pos.keys()) + 1
    prec, rec, scores, pious = [None] * n_cls, [None] * n_cls, [None] * n_cls, [None] * n_cls
    for l in n_pos.keys():
        s = np.array(score[l])
        if s.shape[0] == 0:
            continue
        order = s.argsort()[::-1]
        m = np.array(match[l], dtype=np.int8)[order]
        scores[l], pious[l] = s[order], np.array(predious[l], dtype=np.float64)[order]
        tp, fp = np.cumsum(m == 1), np.cumsum(m == 0)
        prec[l] = tp / (fp + tp)
        with np.errstate(divide="ignore", invalid="ignore"):
            rec[l] = tp / n_pos[l]
    return prec, rec, scores, pious, dict(per_building)


def cal_mious(predious_per_building, iou_thresh, n_cls):
    """suncg_eval.py:968-981 (cal_mious): per class, the mean over buildings of the mean IoU of that building's
    predictions whose best IoU exceeds `iou_thresh`.  predious_per_building: {label: [array per building]}.
    -> list [n_cls] (index 0 and classes without predictions: nan)."""
    mious = [np.nan] * n_cls
    for l, per in predious_per_building.items():
        vals = []
        for arr in per:
            arr = np.asarray(arr, dtype=np.float64)
            mask = arr > iou_thresh
            vals.append(np.mean(arr[mask]) if mask.any() else np.nan)      # the reference takes the mean of an empty slice (nan)
        if vals and 0 < l < n_cls:
            mious[l] = float(np.mean(vals))
    return mious


def calc_ap(prec, rec, use_07_metric=True, scores=None, predious=None):
    """suncg_eval.py:884-965.  -> ap [n_cls] (slot 0 = mean over the classes) and, when `scores` / `predious` are
    given (VOC07 metric), the reference's recall-precision-score-IoU table [n_cls, 11, 4]: at recall t = 0, 0.1 .. 1
    the best precision, the score threshold that reaches it and the best matched IoU (the 'mIoU' rows of its reports)."""
    n = len(prec)
    ap = np.empty(n)
    table = np.full((n, 11, 4), np.nan)
    for l in range(n):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        if use_07_metric:
            ap[l] = 0
            for j, t in enumerate(np.arange(0.0, 1.1, 0.1)):
                reach = rec[l] >= t
                p = 0 if np.sum(reach) == 0 else np.max(np.nan_to_num(prec[l])[reach])
                ap[l] += p / 11
                if scores is not None and predious is not None and scores[l] is not None:
                    iou = 0 if np.sum(reach) == 0 else np.max(np.nan_to_num(predious[l])[reach])
                    below = rec[l] <= t
                    sc = np.max(scores[l]) + 0.01 if np.sum(below) == 0 else np.min(scores[l][below])
                    table[l, j] = [t, p, sc, iou]
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    ap[0] = np.nanmean(ap[1:]) if n > 1 else np.nan                          # class 0 slot = average (:964-965)
    if n > 1:
        with np.errstate(all="ignore"):
            table[0] = np.nanmean(table[1:], axis=0)
    if scores is None or predious is None:
        return ap
    return ap, table


def eval_detection_suncg(preds, gts, cfg, use_07_metric=True, iou_fn=_gpu_iou):
    """Returns {'ap': per-class AP (index 0 = mean), 'map': mean, 'recall_precision_score_iou': table} at cfg.TEST.IOU_THRESHOLD with
    TEST.EVAL_AUG_THICKNESS_* (maskrcnn_benchmark/config/defaults.py:318-320)."""
    ay, az = cfg.TEST.EVAL_AUG_THICKNESS_Y_TAR_ANC, cfg.TEST.EVAL_AUG_THICKNESS_Z_TAR_ANC
    aug = {'target_Y': ay[0], 'anchor_Y': ay[1], 'target_Z': az[0], 'anchor_Z': az[1]}
    prec, rec, scores, pious, per_building = calc_prec_rec(preds, gts, cfg.TEST.IOU_THRESHOLD, aug, iou_fn)
    ap, table = calc_ap(prec, rec, use_07_metric, scores, pious)
    # the IoU row the reference prints under AP (performance_str, suncg_eval.py:217,306: the table averaged over its 11
    # recall steps, IoU column) -- the README's AIoU line; index 0 = mean over the classes like `ap`
    with np.errstate(all="ignore"):
        aiou = np.nanmean(table[:, :, 3], axis=1)
    mious = cal_mious(per_building, cfg.TEST.IOU_THRESHOLD, len(prec))
    return {"ap": ap, "map": float(np.nanmean(ap[1:])), "prec": prec, "rec": rec, "aiou": aiou, "mious": mious,
            "recall_precision_score_iou": table}   # [n_cls, 11, 4]

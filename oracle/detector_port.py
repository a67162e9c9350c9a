"""TEST INFRASTRUCTURE ONLY: CPU composition of the detector's inference path out of oracle ops
(sparse path, boxes) and plain torch-CPU dense layers, driven by a state_dict with the reference's
key names.  Used by tests/ as the checker and by bench.py's cpu_baseline leg ("port") only."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import (bn_forward, box_decode, conv_rules, input_forward, input_sites, nbr_conv, roi_align_rotated_3d,
               rotate_nms_3d, rule_conv, sparse_to_dense, subm_nbr)


def roi_levels(boxes_pixels, scales, canonical_size):
    """LevelMapper_3d (modeling/poolers_3d.py:57-69): argmin over the pooler scales of |scale - sqrt(max(dy, dx)) / canonical|
    on boxes already in pixels.  -> int64 [K]"""
    p = np.asarray(boxes_pixels, np.float32)
    size = np.sqrt(p[:, 3:5].max(1))
    dif = np.abs(np.asarray(scales, np.float32)[None] - (size / np.float32(canonical_size))[:, None])
    return dif.argmin(1)


class OracleFPN:
    """FPN_Net forward (SparseConvNet/sparseconvnet/fpn_net.py:140-203), eval mode with
    track_running_stats=False: every BN normalises with the batch mean / unbiased variance
    (sparseconvnet/batchNormalization.py:51-56)."""

    def __init__(self, sd, full_scale, n_scales, fpn_scales_from_top, roi_scales_from_top,
                 rpn_3d_2d_selector, eps=1e-4, leakiness=0.0, prefix=""):
        self.sd = {k[len(prefix):]: v.detach().cpu().numpy() for k, v in sd.items() if k.startswith(prefix)}
        self.full = np.asarray(full_scale)
        self.n_scales = n_scales
        self.fpn, self.roi, self.sel = list(fpn_scales_from_top), list(roi_scales_from_top), list(rpn_3d_2d_selector)
        self.eps, self.leak = eps, leakiness

    def w(self, key):
        w = self.sd[key]
        return w.reshape(w.shape[0], w.shape[2], w.shape[3])

    def bn(self, x, prefix):
        mean = x.mean(0, dtype=np.float64).astype(np.float32)
        var = x.var(0, ddof=1, dtype=np.float64).astype(np.float32)
        out, *_ = bn_forward(x, mean, var, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"],
                             self.eps, 0.0, False, self.leak)
        return out

    def subm(self, x, loc, key, filt=(3, 3, 3)):
        ck = (id(loc), tuple(filt))
        if ck not in self._nbr:
            self._nbr[ck] = subm_nbr(loc, filt)[0]
        return nbr_conv(x, self.w(key), self._nbr[ck])

    def __call__(self, coords, feats):
        self._nbr = {}
        sop, loc0 = input_sites(coords)
        x = input_forward(feats, sop, loc0.shape[0], True)
        x = self.subm(x, loc0, "layers_in.1.weight")
        locs, rules, downs = [loc0], [], []
        for k in range(self.n_scales):
            if k > 0:
                pre = f"m_downs.{k}.0"
                y = self.bn(x, pre + ".0")
                size = self.full // (2 ** k)
                lo, ru = conv_rules(locs[-1], [2, 2, 2], [2, 2, 2], size)
                locs.append(lo)
                rules.append(ru)
                x = rule_conv(y, self.w(pre + ".1.weight"), ru, lo.shape[0])
                blk = f"m_downs.{k}.1.1"
            else:
                blk = "m_downs.0.0.1"
            y = self.bn(x, blk + ".0")
            y = self.subm(y, locs[k], blk + ".1.weight")
            y = self.bn(y, blk + ".2")
            y = self.subm(y, locs[k], blk + ".3.weight")
            x = x + y
            downs.append(x)
        top = self.n_scales - 1
        net = self.subm(downs[top], locs[top], f"m_shortcuts.{top}.weight", (1, 1, 1))
        ups = [net]
        need = max(self.fpn + self.roi)
        for k in range(need):
            j = self.n_scales - 2 - k
            y = self.bn(net, f"m_ups.{k}.0")
            y = rule_conv(y, self.w(f"m_ups.{k}.1.weight"), rules[j], locs[j].shape[0], deconv=True)
            sc = self.subm(downs[j], locs[j], f"m_shortcuts.{j}.weight", (1, 1, 1))
            net = y + sc
            ups.append(self.subm(net, locs[j], f"m_mergeds.{k}.weight"))
        up_locs = [locs[self.n_scales - 1 - i] for i in range(len(ups))]
        maps3d = [(ups[i], up_locs[i]) for i in self.fpn]
        maps2d = []
        for i, (f, lo) in enumerate(maps3d):
            size = self.full // (2 ** (self.n_scales - 1 - self.fpn[i]))
            z = int(size[2])
            if (i + len(maps3d)) in self.sel:
                lo2, ru2 = conv_rules(lo, [1, 1, z], [1, 1, 1], [size[0], size[1], 1])
                maps2d.append((rule_conv(f, self.w(f"convs_pro2d.{i}.weight"), ru2, lo2.shape[0]), lo2))
            else:
                maps2d.append(None)
        allmaps = maps3d + maps2d
        rpn = [allmaps[i] for i in self.sel]
        roi = [(ups[i], up_locs[i], self.full // (2 ** (self.n_scales - 1 - i))) for i in self.roi]
        return rpn, roi


def _lin(sd, name, x):
    w = sd[name + ".weight"].detach().cpu().float()
    b = sd[name + ".bias"].detach().cpu().float()
    return F.linear(x, w.view(w.shape[0], -1), b)


def nms_clamped(boxes, scores, thresh, aug, max_keep, pre_max=2000):
    """boxlist_nms_3d (structures/boxlist_ops_3d.py:14-62): clamp for the IoU only, pre_max 2000."""
    b = np.asarray(boxes, np.float32).copy()
    b[:, 3:5] = np.maximum(b[:, 3:5], aug[0])
    b[:, 5] = np.maximum(b[:, 5], aug[1])
    order = np.argsort(-np.asarray(scores, np.float64), kind="stable")[:pre_max]
    keep = rotate_nms_3d(b[order], np.asarray(scores, np.float32)[order], thresh)[:max_keep]
    return order[keep]


def rois_from_boxes(boxes_pixels):
    """modeling/poolers_3d.py:107-124 (+ BoxList3D.convert('standard'))."""
    b = torch.as_tensor(boxes_pixels, dtype=torch.float32)
    std = b[:, [0, 1, 2, 4, 3, 5, 6]].clone()
    std[:, 2] += b[:, 5] * 0.5
    std[:, 6] += math.pi * 0.5
    std[:, 6] = std[:, 6] - torch.floor(std[:, 6] / math.pi + 0.0) * math.pi
    rois = torch.cat([torch.zeros((b.shape[0], 1)), std], 1)[:, [0, 2, 1, 3, 5, 4, 6, 7]]
    rois[:, -1] *= 180.0 / math.pi
    return rois.numpy()


def class_groups(separate_classes_id, num_input_classes):
    """SeperateClassifier.__init__ (modeling/seperate_classifier.py:23-44): group 0 = the classes that were not separated
    (real background column 0 first); separated group g >= 1 = [its own background column num_input_classes + g - 1] +
    its sorted classes.  -> list of column lists (empty separate_classes_id: one group of every column)."""
    groups = [sorted(int(c) for c in g) for g in separate_classes_id]
    flat = [c for g in groups for c in g]
    assert 0 not in flat
    out = [[c for c in range(num_input_classes) if c not in flat]]
    for i, g in enumerate(groups):
        out.append([num_input_classes + i] + g)
    return out


def group_targets(grouped_classes, boxes, labels):
    """seperate_targets_and_update_labels (seperate_classifier.py:236-287): per group the GT boxes of its classes, class by
    class in the group's column order (one nonzero per class, concatenated), labels renumbered to the position inside
    the group.  -> list of (boxes [m_g, 7], labels int64 [m_g])."""
    boxes, labels = np.asarray(boxes, np.float32), np.asarray(labels, np.int64)
    out = []
    for classes in grouped_classes:
        ids = [np.nonzero(labels == c)[0] for c in classes]
        new = [np.full(len(i), k, np.int64) for k, i in enumerate(ids)]
        ids, new = np.concatenate(ids), np.concatenate(new)
        out.append((boxes[ids], new))
    return out


def matcher(q, high, low, allow_low_quality, yaw_diff=None, yaw_threshold=3.1416 * 0.4):
    """Matcher.__call__ (modeling/matcher.py:58-177) in numpy: q fp32 [M gt, N] -> int64 [N] (gt index, -1 below the low
    threshold, -2 between).  With yaw_diff the qualities of pairs whose yaws differ by >= yaw_threshold are zeroed
    (:50-55); allow_low_quality: every gt keeps the predictions that tie its best quality (:131-147), and would-be
    negatives within 0.05 of a gt's best (and above 0.02) become ignored (:149-158)."""
    q = np.asarray(q, np.float32)
    if yaw_diff is not None and not yaw_threshold > 1.58:
        q = q * (np.abs(np.asarray(yaw_diff, np.float32)) < np.float32(yaw_threshold)).astype(np.float32)
    vals, matches = q.max(0), q.argmax(0).astype(np.int64)          # ties: lowest gt index, as torch.max
    all_matches = matches.copy()
    below = vals < np.float32(low)
    between = (vals >= np.float32(low)) & (vals < np.float32(high))
    matches[below] = -1
    matches[between] = -2
    if allow_low_quality:
        highest = q.max(1)
        upd = np.nonzero(q == highest[:, None])[1]
        matches[upd] = all_matches[upd]
        thr = np.maximum(np.float32(0.02), highest - np.float32(0.05))
        ignore = (q > thr[:, None]).any(0) & (matches == -1)
        matches[ignore] = -2
    return matches


def rpn_labels(cfg, anchors, gt_boxes, return_iou=False):
    """RPNLossComputation.match_targets_to_anchors + prepare_targets (modeling/rpn/loss_3d.py:88-109,178-205):
    IoU criterion 2 with the label thickness clamps, |yaw difference| wrapped to [-pi/2, pi/2), Matcher with low-quality
    matches -> labels fp32 [N]: 1 positive, 0 negative, -1 ignored."""
    from . import boxes_iou_3d
    rpn = cfg.MODEL.RPN
    anchors, gt_boxes = np.asarray(anchors, np.float32), np.asarray(gt_boxes, np.float32)
    if gt_boxes.shape[0] == 0:
        lab = np.zeros(anchors.shape[0], np.float32)
        return (lab, None) if return_iou else lab
    ay, az = rpn.LABEL_AUG_THICKNESS_Y_TAR_ANC, rpn.LABEL_AUG_THICKNESS_Z_TAR_ANC
    aug = dict(target_Y=ay[0], anchor_Y=ay[1], target_Z=az[0], anchor_Z=az[1])
    q = boxes_iou_3d(gt_boxes, anchors, aug, criterion=2)
    d = gt_boxes[:, 6:7] - anchors[None, :, 6]                                   # angle_dif(anchor, target, 0)
    pi = np.float32(math.pi)
    d = np.abs(d - np.floor(d / pi + np.float32(0.5)) * pi)
    m = matcher(q, rpn.FG_IOU_THRESHOLD, rpn.BG_IOU_THRESHOLD, True, d, rpn.YAW_THRESHOLD)
    lab = (m >= 0).astype(np.float32)
    lab[m == -2] = -1
    return (lab, (q, d, m)) if return_iou else lab


class OracleDetector:
    """SparseRCNN inference (modeling/detector/sparse_rcnn.py:37-76) on the CPU."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        s = cfg.SPARSE3D
        self.fpn = OracleFPN(sd, s.VOXEL_FULL_SCALE, len(s.nPlanesFront), cfg.MODEL.RPN.RPN_SCALES_FROM_TOP,
                             cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_FROM_TOP, cfg.MODEL.RPN.RPN_3D_2D_SELECTOR,
                             prefix="backbone.")
        # class groups (3G6c): [] -> one group of all columns
        self.groups = class_groups(getattr(cfg.MODEL, "SEPARATE_CLASSES_ID", []), len(cfg.INPUT.CLASSES))
        self.rpn_groups = (len(self.groups) - 1) * int(cfg.MODEL.SEPARATE_RPN) + 1

    def anchors(self, locs):
        rpn = self.cfg.MODEL.RPN
        out = []
        for size, use_yaw, stride, loc in zip(rpn.ANCHOR_SIZES_3D, rpn.USE_YAWS, rpn.ANCHOR_STRIDE, locs):
            base = np.zeros((len(rpn.YAWS), 7), np.float32)
            for j in range(len(rpn.YAWS)):
                if use_yaw:
                    base[j, 3:6], base[j, 6] = size, rpn.YAWS[j]
                else:
                    base[j, 3:6] = np.float32(size) * np.float32(rpn.RATIOS[j])
            cent = loc[:, :3].astype(np.float32) / np.float32(self.cfg.SPARSE3D.VOXEL_SCALE) * np.asarray(stride, np.float32)
            a = np.zeros((loc.shape[0], len(rpn.YAWS), 7), np.float32)
            a[:, :, :3] = cent[:, None, :]
            out.append((a + base[None]).reshape(-1, 7))
        return np.concatenate(out)

    def rpn_head(self, rpn_maps):
        """RPNHead.forward + cat_scales_obj_reg (rpn/rpn_sparse3d.py:19-77,109-131): -> objectness [n A, G] (sigmoid not
        applied), regression [n A, 7 G], anchors [n A, 7]; rows ordered scale, site, anchor; the head's A*G (A*7*G) output
        channels of a site are read as [A, G] ([A, 7 G])."""
        G = self.rpn_groups
        obj, reg = [], []
        for f, _ in rpn_maps:
            t = F.relu(_lin(self.sd, "rpn.head.conv", torch.from_numpy(f)))
            obj.append(_lin(self.sd, "rpn.head.cls_logits", t).reshape(-1, G))
            reg.append(_lin(self.sd, "rpn.head.bbox_pred", t).reshape(-1, 7 * G))
        return torch.cat(obj), torch.cat(reg), self.anchors([l for _, l in rpn_maps])

    def rpn_select(self, objectness, reg, anchors, topk_idx=None, scores=None):
        """RPNPostProcessor.forward_for_single_feature_map (rpn/inference_3d.py:82-163) for one objectness column:
        sigmoid, top-k, decode, boxlist_nms_3d.  topk_idx: positions to use instead of this side's own top-k (tests hand
        in the device's choice when scores tie); scores: the sigmoid values to select on instead of this side's own
        (libm's expf differs from the device's in the last bit, which reorders near-ties).  -> (proposals, scores, kept positions in the top-k list, top-k idx)"""
        rc = self.cfg.MODEL.RPN
        if scores is None:
            scores = objectness.sigmoid()
        scores = torch.as_tensor(scores)
        k = min(rc.FPN_PRE_NMS_TOP_N_TEST, scores.shape[0])
        if topk_idx is None:
            # inference_3d.py:109 `topk(sorted=True)`: torch leaves the order among equal scores open (sigmoid saturates to
            # exactly 1.0 for large logits); defined here as descending score, lower index first (= the stable argsort of
            # nms_cpu.py:37's -scores), which d3d_rotate_nms_3d / d3d_topk follow
            idx = torch.from_numpy(np.argsort(-scores.numpy().astype(np.float64), kind="stable")[:k].copy())
            sk = scores[idx]
        else:
            idx = torch.as_tensor(topk_idx, dtype=torch.int64)
            sk = scores[idx]
        props = box_decode(reg[idx].numpy(), anchors[idx.numpy()])
        keep = nms_clamped(props, sk.numpy(), rc.NMS_THRESH, rc.NMS_AUG_THICKNESS_Y_Z, rc.FPN_POST_NMS_TOP_N_TEST)
        return props[keep], sk.numpy()[keep], keep, idx.numpy()

    def rpn(self, rpn_maps):
        obj, reg, anchors = self.rpn_head(rpn_maps)
        assert obj.shape[1] == 1
        p, sc, _, _ = self.rpn_select(obj[:, 0], reg, anchors)
        return p, sc

    def rpn_grouped(self, rpn_maps):
        """seperate_rpn_selector (seperate_classifier.py:58-81): the selector once per class group on the group's
        objectness column and 7 regression columns (budgets already scaled by 1.5 / groups, tools/train_net_sparse3d.py:
        247-255).  -> list over groups of (proposals, scores)"""
        obj, reg, anchors = self.rpn_head(rpn_maps)
        assert obj.shape[1] == len(self.groups)
        return [self.rpn_select(obj[:, g], reg[:, 7 * g:7 * g + 7], anchors)[:2] for g in range(obj.shape[1])]

    def pool(self, roi_maps, proposals, batch_ids=None, n_examples=1):
        """batch_ids [K]: example of every proposal (poolers_3d.py:112-118 writes it into column 0 of the RoI)."""
        head = self.cfg.MODEL.ROI_BOX_HEAD
        p = np.asarray(proposals, np.float32).copy()
        p[:, 0:6] *= self.cfg.SPARSE3D.VOXEL_SCALE
        rois = rois_from_boxes(p)
        if batch_ids is not None:
            rois[:, 0] = np.asarray(batch_ids, np.float32)
        levels = roi_levels(p, head.POOLER_SCALES_SPATIAL, head.CANONICAL_SIZE)
        ph, pw, pz = head.POOLER_RESOLUTION
        out = np.zeros((p.shape[0], roi_maps[0][0].shape[1], ph, pw, pz), np.float32)
        for lvl, (f, loc, size3) in enumerate(roi_maps):
            idx = np.nonzero(levels == lvl)[0]
            if not len(idx):
                continue
            crop = loc[:, :3].max(0) + 1                                  # tools_3d_2d.py:16-29
            dense = sparse_to_dense(f, loc, [int(v) for v in crop], n_examples)   # same values as crop of the full map
            out[idx] = roi_align_rotated_3d(dense, rois[idx], head.POOLER_SCALES_SPATIAL[lvl], ph, pw, pz,
                                            head.POOLER_SAMPLING_RATIO)
        return out

    def box_head(self, pooled):
        sd = self.sd
        x = torch.from_numpy(pooled)
        pre = "roi_heads.box.feature_extractor."
        x = F.conv3d(x, sd[pre + "conv3d.0.weight"].cpu().float(), sd[pre + "conv3d.0.bias"].cpu().float())
        x = F.relu(F.batch_norm(x, None, None, sd[pre + "conv3d.1.weight"].cpu().float(),
                                sd[pre + "conv3d.1.bias"].cpu().float(), True, 0.1, 1e-5))
        x = x.reshape(x.shape[0], -1)
        x = F.relu(_lin(sd, pre + "fc6", x))
        x = F.relu(_lin(sd, pre + "fc7", x))
        return _lin(sd, "roi_heads.box.predictor.cls_score", x), _lin(sd, "roi_heads.box.predictor.bbox_pred", x)

    def post(self, logits, reg, proposals, prob=None):
        """prob: the class probabilities to select on instead of this side's own softmax (the device's expf differs from
        libm's in the last bit, which reorders near-tied candidates; tests hand in the device's values)."""
        rh = self.cfg.MODEL.ROI_HEADS
        prob = F.softmax(logits, -1).numpy() if prob is None else np.asarray(prob, np.float32)
        nc = prob.shape[1]
        reg = reg.numpy()
        ob, os_, ol = [], [], []
        for j in range(1, nc):
            inds = np.nonzero(prob[:, j] > rh.SCORE_THRESH)[0]
            if not len(inds):
                continue
            bj = box_decode(reg[inds, 7 * j:7 * j + 7], np.asarray(proposals, np.float32)[inds], rh.BBOX_REG_WEIGHTS)
            keep = nms_clamped(bj, prob[inds, j], rh.NMS, rh.NMS_AUG_THICKNESS_Y_Z, 500)
            ob.append(bj[keep]); os_.append(prob[inds, j][keep]); ol.append(np.full(len(keep), j))
        if not ob:
            return np.zeros((0, 7), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64)
        b, s, l = np.concatenate(ob), np.concatenate(os_), np.concatenate(ol)
        if len(s) > rh.DETECTIONS_PER_IMG > 0:
            th = np.sort(s)[len(s) - rh.DETECTIONS_PER_IMG]
            k = s >= th
            b, s, l = b[k], s[k], l[k]
        return b, s, l

    def post_grouped(self, logits, reg, proposals, sep_id, probs=None):
        """SeperateClassifier.post_processor (seperate_classifier.py:299-321): rows of group g restricted to the group's
        class columns (:217-234), the plain post-processor per group (DETECTIONS_PER_IMG already scaled), labels mapped
        back to the original class ids, groups concatenated in order.  probs: per group the probabilities of its rows (see
        `post`).  -> (boxes, scores, labels)"""
        sep_id = np.asarray(sep_id)
        n = logits.shape[0]
        ob, os_, ol = [], [], []
        for g, cols in enumerate(self.groups):
            ids = torch.from_numpy(np.nonzero(sep_id == g)[0])
            c = torch.tensor(cols)
            lg = logits[ids][:, c]
            rg = reg.view(n, -1, 7)[:, c, :].reshape(n, -1)[ids]
            b, sc, lab = self.post(lg, rg, np.asarray(proposals, np.float32)[ids.numpy()],
                                   None if probs is None else probs[g])
            ob.append(b); os_.append(sc); ol.append(np.asarray(cols, np.int64)[lab])
        return np.concatenate(ob), np.concatenate(os_), np.concatenate(ol)

    def __call__(self, coords, feats):
        rpn_maps, roi_maps = self.fpn(coords, feats)
        if len(self.groups) > 1 and self.rpn_groups > 1:
            per = self.rpn_grouped(rpn_maps)
            props = np.concatenate([p for p, _ in per]).copy()
            sep_id = np.concatenate([np.full(len(p), g, np.int64) for g, (p, _) in enumerate(per)])
            props[:, 3:6] = np.maximum(props[:, 3:6], 0.001)
            logits, reg = self.box_head(self.pool(roi_maps, props))
            return self.post_grouped(logits, reg, props, sep_id)
        props, _ = self.rpn(rpn_maps)
        props = props.copy()
        props[:, 3:6] = np.maximum(props[:, 3:6], 0.001)
        logits, reg = self.box_head(self.pool(roi_maps, props))
        return self.post(logits, reg, props)

    def detect_batch(self, coords, feats, n_examples):
        """Several examples per batch (coords [N,4], examples listed one after the other): ONE backbone pass (BatchNorm
        over all rows, as the reference), RPN selection per example (rpn/inference_3d.py:92-163), one box-head pass over
        the RoIs of all examples, post-processing per example (box_head_3d/inference.py:66-99).
        -> (list over examples of (boxes, scores, labels), intermediates)"""
        rpn_maps, roi_maps = self.fpn(coords, feats)
        props, ids = [], []
        for b in range(n_examples):
            maps_b = [(f[loc[:, 3] == b], loc[loc[:, 3] == b]) for f, loc in rpn_maps]
            p, _ = self.rpn(maps_b)
            p = p.copy()
            p[:, 3:6] = np.maximum(p[:, 3:6], 0.001)
            props.append(p)
            ids.append(np.full(len(p), b, np.int32))
        allp, allid = np.concatenate(props), np.concatenate(ids)
        logits, reg = self.box_head(self.pool(roi_maps, allp, allid, n_examples))
        out = []
        for b in range(n_examples):
            m = torch.from_numpy(allid == b)
            out.append(self.post(logits[m], reg[m], allp[allid == b]))
        return out, {"proposals": props, "logits": logits, "reg": reg, "rpn_maps": rpn_maps, "roi_maps": roi_maps}

"""Inference path of the reference's SparseRCNN (maskrcnn_benchmark/modeling/detector/sparse_rcnn.py:37-76)
for the non-separated configs (4c / 6c): backbone -> RPN head -> anchors -> top-k + decode + rotated NMS
-> rotated 3-D RoIAlign -> box head -> per-class rotated NMS -> top detections.

Module / parameter names follow the reference so that its checkpoints load unchanged
(`backbone.*`, `rpn.head.{conv,cls_logits,bbox_pred}.*`,
`roi_heads.box.feature_extractor.{conv3d.0,conv3d.1,fc6,fc7}.*`, `roi_heads.box.predictor.*`).
Dense per-site / per-RoI linear algebra (1x1 convs, fc layers) runs on rocBLAS / MIOpen through
PyTorch; everything sparse, geometric or combinatorial runs in libd3d_hip.so.

Boxes are plain tensors [n,7] in yx_zb mode (xc, yc, z_bot, dy, dx, dz, yaw) instead of BoxList3D
objects; fields travel next to them in a small dict.  SEPARATE_CLASSES (3G6c) runs the RPN selection / losses
and the RoI losses / post-processing once per class group (SeperateClassifier).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from ._lib import check, lib, ptr, stream_of
from . import sparseconvnet as scn
from . import training as T
from .config import class_to_label
from .roi_align_rotated_3d import roi_align_rotated_3d_sparse, roi_align_rotated_3d_sparse_into, roi_prepare
from .sparseconvnet import SCN


class SeperateClassifier(object):
    """Class grouping of modeling/seperate_classifier.py:7-55 (3G6c): group 0 = the classes that were not
    separated (with the real background column 0); every separated group g >= 1 gets its own background column
    num_classes + g - 1 followed by its (sorted) classes."""

    def __init__(self, separate_classes_id, num_input_classes):
        groups = [sorted(g) for g in separate_classes_id]
        self.need_seperate = len(groups) > 0
        self.num_input_classes = num_input_classes
        flat = [c for g in groups for c in g]
        assert 0 not in flat
        self.grouped_classes = [[c for c in range(num_input_classes) if c not in flat]]
        for i, g in enumerate(groups):
            self.grouped_classes.append([num_input_classes + i] + g)
        self.group_num = len(self.grouped_classes)
        self.total_classes = self.seperated_num_classes_total = num_input_classes + self.group_num - 1
        self.class_nums = [len(g) for g in self.grouped_classes]
        # original label -> (group, label inside the group) and back (:39-44)
        self.org_labels_to_sep_labels = torch.full((self.total_classes, 2), -1, dtype=torch.int64)
        self.sep_labels_to_org_labels = [torch.tensor(g, dtype=torch.int64) for g in self.grouped_classes]
        for g, classes in enumerate(self.grouped_classes):
            for i, c in enumerate(classes):
                self.org_labels_to_sep_labels[c] = torch.tensor([g, i])

    def group_targets(self, targets):
        """seperate_targets_and_update_labels (:268-297) -> per group {"bbox3d", "labels"}: the group's boxes in class
        order (one nonzero per class, concatenated) with labels renumbered inside the group."""
        out = []
        for classes in self.grouped_classes:
            lab = targets["labels"]
            sel, new = [], []
            for i, c in enumerate(classes):
                ids = torch.nonzero(lab == c).view(-1)
                sel.append(ids)
                new.append(torch.full_like(ids, i))
            sel, new = torch.cat(sel), torch.cat(new)
            out.append({"bbox3d": targets["bbox3d"][sel], "labels": new})
        return out

    def seperate_pred_logits(self, class_logits, sep_ids_g):
        """:222-229: rows of group g (sep_ids_g[g]) restricted to the group's columns."""
        assert class_logits.shape[1] == self.total_classes
        return [class_logits[ids][:, torch.tensor(cols, device=class_logits.device)]
                for ids, cols in zip(sep_ids_g, self.grouped_classes)]

    def seperate_pred_box(self, box_regression, sep_ids_g):
        """:231-239: the 7-column blocks of the group's classes."""
        n = box_regression.shape[0]
        assert box_regression.shape[1] == self.total_classes * 7
        return [box_regression.view(n, -1, 7)[:, torch.tensor(cols, device=box_regression.device), :].reshape(n, -1)[ids]
                for ids, cols in zip(sep_ids_g, self.grouped_classes)]

    def org_label(self, gi, group_labels):
        return self.sep_labels_to_org_labels[gi].to(group_labels.device)[group_labels]


def build_backbone(cfg):
    """maskrcnn_benchmark/modeling/backbone/backbone.py:38-69 ("Sparse-R-50-FPN")."""
    s = cfg.SPARSE3D
    return scn.FPN_Net(s.VOXEL_FULL_SCALE, 3, cfg.INPUT.ELEMENTS, s.BLOCK_REPS, s.nPlanesFront,
                       nPlaneM=s.nPlaneMap, residual_blocks=s.RESIDUAL_BLOCK,
                       fpn_scales_from_top=cfg.MODEL.RPN.RPN_SCALES_FROM_TOP,
                       roi_scales_from_top=cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_FROM_TOP,
                       downsample=[s.KERNEL, s.STRIDE], rpn_map_sizes=cfg.MODEL.RPN.RPN_MAP_SIZES,
                       voxel_scale=s.VOXEL_SCALE, rpn_3d_2d_selector=cfg.MODEL.RPN.RPN_3D_2D_SELECTOR,
                       bn_momentum=cfg.SOLVER.BN_MOMENTUM, track_running_stats=cfg.SOLVER.TRACK_RUNNING_STATS)


# ----------------------------------------------------------------------------------------------
class AnchorGenerator(nn.Module):
    """modeling/rpn/anchor_generator_sparse3d.py:44-120,207-241: one anchor size per selected map,
    4 anchors per active site (4 yaws, or 4 size ratios with yaw 0)."""

    def __init__(self, cfg):
        super().__init__()
        rpn = cfg.MODEL.RPN
        yaws = np.array(rpn.YAWS, dtype=np.float32).reshape(-1, 1)
        ratios = np.array(rpn.RATIOS, dtype=np.float32)
        cells = []
        for size, use_yaw in zip(np.array(rpn.ANCHOR_SIZES_3D, dtype=np.float32), rpn.USE_YAWS):
            rows = []
            for j in range(yaws.shape[0]):
                if use_yaw:
                    rows.append(np.concatenate([np.zeros(3, np.float32), size, yaws[j]]))
                else:
                    rows.append(np.concatenate([np.zeros(3, np.float32), size * ratios[j], np.zeros(1, np.float32)]))
            cells.append(torch.from_numpy(np.stack(rows).astype(np.float32)))
        self.cell_anchors = cells
        self.strides = torch.tensor(np.array(rpn.ANCHOR_STRIDE, dtype=np.float32))
        self.voxel_scale = cfg.SPARSE3D.VOXEL_SCALE
        self.anchor_num_per_loc = yaws.shape[0]

    def num_anchors_per_location(self):
        return self.anchor_num_per_loc

    def forward_cat(self, feature_maps_sparse):
        """torch.cat(self(feature_maps_sparse), 0) written in place by one d3d_anchors launch per map."""
        A = self.anchor_num_per_loc
        ns = [f.features.shape[0] for f in feature_maps_sparse]
        out = torch.empty((sum(ns) * A, 7), dtype=torch.float32, device=feature_maps_sparse[0].features.device)
        o = 0
        for base, fmap, stride, n in zip(self.cell_anchors, feature_maps_sparse, self.strides, ns):
            if n:
                fmap.metadata.anchors(fmap.spatial_size, base.tolist(), stride.tolist(), self.voxel_scale,
                                      out[o * A:(o + n) * A])
            o += n
        return out

    def forward(self, feature_maps_sparse):
        anchors = []
        for base, fmap, stride in zip(self.cell_anchors, feature_maps_sparse, self.strides):
            loc = fmap.get_spatial_locations()
            dev = loc.device
            # :99.  The reference evaluates this on the CPU (true division); a GPU tensor divisor keeps
            # torch from rewriting x / 50 as x * (1 / 50), which differs in the last bit.
            vs = torch.full((1, 1), float(self.voxel_scale), dtype=torch.float32, device=dev)
            cent = loc[:, 0:3].float() / vs * stride.to(dev).view(1, 3)
            cent = torch.cat([cent, torch.zeros(cent.shape[0], 4, device=dev)], 1).view(-1, 1, 7)
            anchors.append((cent + base.to(dev).view(1, -1, 7)).reshape(-1, 7))
        return anchors


class RPNHead(nn.Module):
    """modeling/rpn/rpn_sparse3d.py:80-131 (SingleConvRPNHead_Sparse3D): three 1x1 convolutions over
    the active sites, i.e. per-site linear layers.  Parameters keep the Conv2d shapes [out,in,1,1]."""

    def __init__(self, cfg, in_channels, num_anchors_per_location):
        super().__init__()
        self.num_anchors_per_location = num_anchors_per_location
        self.seperate_rpn = int(len(cfg.MODEL.SEPARATE_CLASSES) * cfg.MODEL.SEPARATE_RPN) + 1
        a = num_anchors_per_location * self.seperate_rpn
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.cls_logits = nn.Conv2d(in_channels, a, kernel_size=1)
        self.bbox_pred = nn.Conv2d(in_channels, a * 7, kernel_size=1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    @staticmethod
    def _lin(layer, x):
        return F.linear(x, layer.weight.view(layer.weight.shape[0], -1), layer.bias)

    def forward(self, features):
        """features: list of [n_s, C]  ->  objectness [sum n_s*A], regression [sum n_s*A, 7] in the
        flattening order of cat_scales_obj_reg (:19-77): scale, site, anchor."""
        if not torch.is_grad_enabled() and len(features) > 1:
            # the head is shared by the scales and acts per site: one GEMM over the concatenated sites
            t = F.relu(self._lin(self.conv, torch.cat(features, 0)))
            return (self._lin(self.cls_logits, t).reshape(-1, self.seperate_rpn),
                    self._lin(self.bbox_pred, t).reshape(-1, 7 * self.seperate_rpn))
        obj, reg = [], []
        for f in features:
            t = F.relu(self._lin(self.conv, f))
            obj.append(self._lin(self.cls_logits, t).reshape(-1, self.seperate_rpn))
            reg.append(self._lin(self.bbox_pred, t).reshape(-1, 7 * self.seperate_rpn))
        return torch.cat(obj, 0), torch.cat(reg, 0)


class RPNModule(nn.Module):
    """modeling/rpn/rpn_sparse3d.py:137-231 + rpn/inference_3d.py:82-163 (test path)."""

    def __init__(self, cfg):
        super().__init__()
        self.anchor_generator = AnchorGenerator(cfg)
        self.head = RPNHead(cfg, cfg.MODEL.BACKBONE.OUT_CHANNELS, self.anchor_generator.num_anchors_per_location())
        rpn = cfg.MODEL.RPN
        self.top_n = {False: (rpn.FPN_PRE_NMS_TOP_N_TEST, rpn.FPN_POST_NMS_TOP_N_TEST),
                      True: (rpn.FPN_PRE_NMS_TOP_N_TRAIN, rpn.FPN_POST_NMS_TOP_N_TRAIN)}
        self.nms_thresh = rpn.NMS_THRESH
        self.nms_aug_thickness = list(rpn.NMS_AUG_THICKNESS_Y_Z)
        self.add_gt_proposals = rpn.ADD_GT_PROPOSALS
        self.loss_evaluator = T.RPNLoss(cfg)
        self.sep = SeperateClassifier(cfg.MODEL.SEPARATE_CLASSES_ID, len(cfg.INPUT.CLASSES))

    @torch.no_grad()
    def select_proposals(self, objectness, box_regression, anchors, train):
        pre, post = self.top_n[bool(train)]
        scores = objectness.reshape(-1).sigmoid()
        k = min(pre, scores.shape[0])
        scores_k, idx = scores.topk(k, dim=0, sorted=True)                      # inference_3d.py:109
        proposals = box_ops.box_decode(box_regression[idx], anchors[idx])       # :123
        keep = box_ops.nms_3d_presorted(proposals, self.nms_thresh, self.nms_aug_thickness, max_proposals=post,
                                        flag='rpn_post')                        # scores_k is sorted: no re-sort
        return proposals[keep], scores_k[keep]

    def forward(self, features_sparse, targets=None):
        """eval: (proposals, objectness).  train: (proposals incl. GT boxes, objectness, loss dict)
        (rpn_sparse3d.py:233-270, rpn/inference_3d.py:53-80,180-199)."""
        objectness, box_regression = self.head([f.features for f in features_sparse])
        with torch.no_grad():
            anchors = self.anchor_generator.forward_cat(features_sparse)
        assert objectness.shape[0] == box_regression.shape[0] == anchors.shape[0]
        if self.sep.need_seperate and self.head.seperate_rpn > 1:
            return self._forward_grouped(anchors, objectness, box_regression, targets)
        proposals, scores = self.select_proposals(objectness.detach(), box_regression.detach(), anchors,
                                                  self.training)
        if not self.training:
            return proposals, scores
        gt = targets["bbox3d"]
        if self.add_gt_proposals and gt.shape[0]:
            proposals = torch.cat([proposals, gt], 0)
            scores = torch.cat([scores, torch.ones(gt.shape[0], device=scores.device)], 0)
        loss_obj, loss_reg = self.loss_evaluator(anchors, objectness.reshape(-1), box_regression, gt)
        return proposals, scores, {"loss_objectness": loss_obj, "loss_rpn_box_reg": loss_reg}

    def _forward_grouped(self, anchors, objectness, box_regression, targets):
        """seperate_rpn_selector / seperate_rpn_loss_evaluator (seperate_classifier.py:58-95): one proposal set
        and one loss pair per class group; proposals carry their group id (`sep_id`)."""
        props, scores, sep_ids, losses = [], [], [], {}
        tg = self.sep.group_targets(targets) if self.training else [None] * self.sep.group_num
        for gi in range(self.sep.group_num):
            obj_g, reg_g = objectness[:, gi], box_regression[:, 7 * gi:7 * gi + 7]
            p, sc = self.select_proposals(obj_g.detach(), reg_g.detach().contiguous(), anchors, self.training)
            if self.training:
                gt = tg[gi]["bbox3d"]
                if self.add_gt_proposals and gt.shape[0]:
                    p = torch.cat([p, gt], 0)
                    sc = torch.cat([sc, torch.ones(gt.shape[0], device=sc.device)], 0)
                lo, lr = self.loss_evaluator(anchors, obj_g, reg_g, gt)
                losses[f"loss_objectness_{gi}"], losses[f"loss_rpn_box_reg_{gi}"] = lo, lr
            props.append(p)
            scores.append(sc)
            sep_ids.append(torch.full((p.shape[0],), gi, dtype=torch.int64, device=p.device))
        out = (torch.cat(props), torch.cat(scores), torch.cat(sep_ids))
        return out + (losses,) if self.training else out


# ----------------------------------------------------------------------------------------------
def convert_to_roi_format(boxes_yxzb):
    """modeling/poolers_3d.py:107-124 with BoxList3D.convert('standard')
    (structures/bounding_box_3d.py:221-242, limit_yaw in the constructor :167): batch id 0."""
    b = boxes_yxzb
    std = b[:, [0, 1, 2, 4, 3, 5, 6]].clone()
    std[:, 2] += b[:, 5] * 0.5
    std[:, 6] += math.pi * 0.5
    std[:, 6] = box_ops.limit_period(std[:, 6], 0.0, math.pi)
    rois = torch.cat([torch.zeros((b.shape[0], 1), dtype=b.dtype, device=b.device), std], 1)
    rois = rois[:, [0, 2, 1, 3, 5, 4, 6, 7]]
    rois[:, -1] *= 180.0 / math.pi
    return rois


class Pooler(nn.Module):
    """modeling/poolers_3d.py:57-69,73-168 on sparse maps (no dense intermediate)."""

    def __init__(self, output_size, scales, sampling_ratio, canonical_size):
        super().__init__()
        self.output_size, self.scales = tuple(output_size), list(scales)
        self.sampling_ratio, self.canonical_size = sampling_ratio, canonical_size

    def map_levels(self, boxes):
        size = torch.sqrt(boxes[:, 3:5].max(dim=1)[0])
        rate = size / self.canonical_size
        dif = torch.abs(torch.tensor(self.scales, device=boxes.device)[None, :] - rate[:, None])
        return torch.argmin(dif, 1)

    def pool_metric(self, x, boxes_metric, voxel_scale, channels_inner=True):
        """Inference: metric proposals -> pooled features with ONE pre-processing launch (pixels, RoI format and FPN
        level: d3d_roi_prepare, bit-identical to convert_to_roi_format / map_levels) and one launch per level that
        fills its RoIs' slots of the result in place (no nonzero / index_put, no host synchronisation)."""
        ph, pw, pz = self.output_size
        rois, levels = roi_prepare(boxes_metric, voxel_scale, self.scales, self.canonical_size)
        K, C = rois.shape[0], x[0].features.shape[1]
        out = torch.empty((K, ph, pw, C, pz) if channels_inner else (K, C, ph, pw, pz), dtype=torch.float32,
                          device=rois.device)
        for level, (fmap, scale) in enumerate(zip(x, self.scales)):       # crop = occupied extent, found on the device
            roi_align_rotated_3d_sparse_into(out, fmap, rois, scale, self.sampling_ratio, crop=None,
                                             roi_levels=levels, level=level, channels_inner=channels_inner)
        return out

    def forward(self, x, boxes_pixels, channels_inner=False):
        """-> [K, C, ph, pw, pz]; channels_inner (inference only): [K, ph, pw, C, pz]."""
        with torch.no_grad():
            rois = convert_to_roi_format(boxes_pixels)
        ph, pw, pz = self.output_size
        if not torch.is_grad_enabled():
            # one result tensor filled in place by one launch per level (no nonzero / index_put, no host sync)
            K, C = rois.shape[0], x[0].features.shape[1]
            out = torch.empty((K, ph, pw, C, pz) if channels_inner else (K, C, ph, pw, pz), dtype=torch.float32,
                              device=rois.device)
            levels = self.map_levels(boxes_pixels).to(torch.int32) if len(self.scales) > 1 else None
            for level, (fmap, scale) in enumerate(zip(x, self.scales)):   # crop = occupied extent, found on the device
                roi_align_rotated_3d_sparse_into(out, fmap, rois, scale, self.sampling_ratio, crop=None,
                                                 roi_levels=levels, level=level, channels_inner=channels_inner)
            return out
        assert not channels_inner
        if len(self.scales) == 1:
            return roi_align_rotated_3d_sparse(x[0], rois, self.scales[0], ph, pw, pz, self.sampling_ratio)
        levels = self.map_levels(boxes_pixels)
        result = torch.zeros((rois.shape[0], x[0].features.shape[1], ph, pw, pz), dtype=torch.float32,
                             device=rois.device)
        for level, (fmap, scale) in enumerate(zip(x, self.scales)):
            idx = torch.nonzero(levels == level).squeeze(1)
            if idx.numel():
                result[idx] = roi_align_rotated_3d_sparse(fmap, rois[idx].contiguous(), scale, ph, pw, pz,
                                                          self.sampling_ratio)
        return result


class FPN2MLPFeatureExtractor(nn.Module):
    """roi_heads/box_head_3d/roi_box_feature_extractors.py:47-117."""

    def __init__(self, cfg):
        super().__init__()
        head = cfg.MODEL.ROI_BOX_HEAD
        res = head.POOLER_RESOLUTION
        self.pooler = Pooler(res, head.POOLER_SCALES_SPATIAL, head.POOLER_SAMPLING_RATIO, head.CANONICAL_SIZE)
        self.voxel_scale = cfg.SPARSE3D.VOXEL_SCALE
        c, rep = cfg.MODEL.BACKBONE.OUT_CHANNELS, head.MLP_HEAD_DIM
        self.conv3d = nn.Sequential(nn.Conv3d(c, rep, kernel_size=[1, 1, res[2]], stride=[1, 1, 1]),
                                    nn.BatchNorm3d(rep, track_running_stats=cfg.SOLVER.TRACK_RUNNING_STATS),
                                    nn.ReLU(inplace=True))
        self.fc6 = nn.Linear(c * res[0] * res[1] * res[2], rep)
        self.fc7 = nn.Linear(rep, rep)
        for l in (self.fc6, self.fc7):
            nn.init.kaiming_uniform_(l.weight, a=1)
            nn.init.constant_(l.bias, 0)

    def _head_conv(self, pooled):
        """conv3d (kernel [1,1,pz] over a z extent of exactly pz) + BatchNorm3d + ReLU.  The convolution
        is one [K*ph*pw, C*pz] x [C*pz, rep] GEMM (rocBLAS) instead of a MIOpen conv3d search."""
        conv, bn, relu = self.conv3d[0], self.conv3d[1], self.conv3d[2]
        K, C, ph, pw, pz = pooled.shape
        if tuple(conv.kernel_size) == (1, 1, pz) and tuple(conv.stride) == (1, 1, 1):
            a = pooled.permute(0, 2, 3, 1, 4).reshape(K * ph * pw, C * pz)
            y = torch.addmm(conv.bias, a, conv.weight.view(conv.out_channels, C * pz).t())
            y = y.view(K, ph, pw, conv.out_channels, 1).permute(0, 3, 1, 2, 4)
            return relu(bn(y.contiguous()))
        return self.conv3d(pooled)

    def _fc6_rows_weight(self, cells):
        """fc6.weight with its input index reordered from (channel, cell) to (cell, channel): the operand for the
        row-major [K, cells, rep] activations of the inference path (cached per weight version)."""
        w = self.fc6.weight
        key = (w.data_ptr(), w._version, cells)
        if getattr(self, "_fc6_rows", (None,))[0] != key:
            rep = w.shape[1] // cells
            self._fc6_rows = (key, w.detach().view(w.shape[0], rep, cells).permute(0, 2, 1).reshape(w.shape[0], -1)
                              .contiguous())
        return self._fc6_rows[1]

    def _forward_rows(self, x0, p):
        """Inference path without layout changes (p: metric proposals): the pooler writes [K, ph, pw, C, pz], whose rows feed the
        [1,1,pz] convolution as a GEMM; BatchNorm3d + ReLU is one row-wise BatchNorm over [K*ph*pw, rep]
        (batch statistics, biased variance: F.batch_norm in training mode), fc6 reads the rows in place."""
        conv, bn = self.conv3d[0], self.conv3d[1]
        pooled = self.pooler.pool_metric(x0, p, self.voxel_scale, channels_inner=True)
        K, ph, pw, C, pz = pooled.shape
        y = torch.addmm(conv.bias, pooled.view(K * ph * pw, C * pz), conv.weight.view(conv.out_channels, C * pz).t())
        rep = y.shape[1]
        out, sm, si = y.new_empty(0), y.new_empty(rep), y.new_empty(rep)
        SCN.BatchNormalization_updateOutput(y, out, sm, si, y.new_zeros(rep), y.new_ones(rep), bn.weight, bn.bias,
                                            bn.eps, 0.0, True, 0.0)
        h = torch.addmm(self.fc6.bias, out.view(K, ph * pw * rep), self._fc6_rows_weight(ph * pw).t())
        return F.relu(self.fc7(F.relu(h)))

    def forward(self, x0, proposals):
        conv, bn = self.conv3d[0], self.conv3d[1]
        if (not torch.is_grad_enabled() and tuple(conv.kernel_size) == (1, 1, self.pooler.output_size[2])
                and tuple(conv.stride) == (1, 1, 1) and (bn.training or not bn.track_running_stats)
                and proposals.shape[0] > 0):
            return self._forward_rows(x0, proposals)                            # metric boxes: pixels on the device
        p = proposals.clone()
        p[:, 0:6] *= self.voxel_scale                                           # convert_metric_to_pixel
        x1 = self._head_conv(self.pooler(x0, p))
        x2 = x1.reshape(x1.size(0), -1)
        return F.relu(self.fc7(F.relu(self.fc6(x2))))


class FPNPredictor(nn.Module):
    """roi_heads/box_head_3d/roi_box_predictors.py:33-55."""

    def __init__(self, cfg):
        super().__init__()
        nc = len(cfg.INPUT.CLASSES) + len(cfg.MODEL.SEPARATE_CLASSES)
        rep = cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
        self.cls_score = nn.Linear(rep, nc)
        self.bbox_pred = nn.Linear(rep, nc * 7)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in (self.cls_score, self.bbox_pred):
            nn.init.constant_(l.bias, 0)

    def forward(self, x):
        return self.cls_score(x), self.bbox_pred(x)


class PostProcessor(nn.Module):
    """roi_heads/box_head_3d/inference.py:40-149."""

    def __init__(self, cfg):
        super().__init__()
        rh = cfg.MODEL.ROI_HEADS
        self.score_thresh, self.nms = rh.SCORE_THRESH, rh.NMS
        self.nms_aug_thickness = list(rh.NMS_AUG_THICKNESS_Y_Z)
        self.detections_per_img = rh.DETECTIONS_PER_IMG
        self.weights = tuple(rh.BBOX_REG_WEIGHTS)

    @torch.no_grad()
    def forward(self, class_logits, box_regression, proposals):
        prob = F.softmax(class_logits, -1)
        boxes = box_ops.box_decode(box_regression, proposals, self.weights)     # [K, 7*nc]
        nc = prob.shape[1]
        K = prob.shape[0]
        if nc < 2 or K == 0:
            z = prob.new_zeros
            return {"bbox3d": z((0, 7)), "scores": z((0,)), "labels": torch.zeros(0, dtype=torch.int64, device=prob.device)}
        # the per-class loop of inference.py:113-139 as ONE batched NMS: class j's candidates (prob > thresh) in
        # descending score order (ties: lower RoI first) are a segment of box indices roi*nc + j; the glue around the
        # sort, the NMS and the top-k is three library launches (d3d_post_*), bit-identical to _select_reference
        dev, prob = prob.device, prob.contiguous()
        sc = torch.empty((nc - 1, K), dtype=torch.float32, device=dev)
        counts = torch.empty((nc - 1,), dtype=torch.int32, device=dev)
        check(lib().d3d_post_scores(ptr(prob), K, nc, float(self.score_thresh), ptr(sc), ptr(counts), stream_of()))
        idx = torch.sort(sc, dim=1, descending=True, stable=True)[1]
        order = torch.empty((nc - 1, K), dtype=torch.int32, device=dev)
        check(lib().d3d_post_order(ptr(idx), K, nc, ptr(order), stream_of()))
        n_max = min(K, 2000)                                                     # pre_max_size of rotate_nms_3d
        keep, nk = box_ops.nms_3d_batched(boxes.view(-1, 7), order, counts, n_max, self.nms, self.nms_aug_thickness,
                                          500)                                   # post_max_size (boxlist_ops_3d.py)
        s_all = torch.empty(((nc - 1) * n_max,), dtype=torch.float32, device=dev)
        flat_all = torch.empty(((nc - 1) * n_max,), dtype=torch.int64, device=dev)
        check(lib().d3d_post_gather(ptr(keep), ptr(nk), nc - 1, n_max, ptr(prob), ptr(s_all), ptr(flat_all), stream_of()))
        if 0 < self.detections_per_img < s_all.shape[0]:                         # :140-148 without a second read-back:
            # the D-th largest score; with fewer than D survivors it is a padding entry (-1) and every survivor stays
            thresh = torch.topk(s_all, self.detections_per_img, sorted=True)[0][-1]
            sel = s_all >= thresh.clamp_min(0.0)                                 # survivors have scores > 0, padding -1
        else:
            sel = s_all >= 0.0
        flat = flat_all[sel]                                                     # the one host synchronisation
        b, s, l = boxes.view(-1, 7)[flat], prob.reshape(-1)[flat], flat % nc
        return {"bbox3d": b, "scores": s, "labels": l}

    def _select_reference(self, prob, boxes):
        """The same selection spelled in tensor ops (what forward ran before the d3d_post_* launches; kept as the
        parity reference of tests/test_detector_gpu.py)."""
        K, nc = prob.shape
        sc = prob[:, 1:].t().contiguous()                                        # [nc-1, K]
        cand = sc > self.score_thresh
        idx = torch.sort(torch.where(cand, sc, sc.new_full((), -1.0)), dim=1, descending=True, stable=True)[1]
        order = (idx * nc + torch.arange(1, nc, device=prob.device).view(-1, 1)).to(torch.int32).contiguous()
        counts = cand.sum(1).to(torch.int32)
        n_max = min(K, 2000)
        keep, nk = box_ops.nms_3d_batched(boxes.view(-1, 7), order, counts, n_max, self.nms, self.nms_aug_thickness, 500)
        valid = torch.arange(keep.shape[1], device=prob.device).view(1, -1) < nk.view(-1, 1)
        flat_all = torch.where(valid, keep, torch.zeros_like(keep)).long().view(-1)   # class-major, selection order
        s_all = torch.where(valid.view(-1), prob.reshape(-1)[flat_all], prob.new_full((), -1.0))
        if 0 < self.detections_per_img < s_all.shape[0]:
            thresh = torch.topk(s_all, self.detections_per_img, sorted=True)[0][-1]
            sel = valid.view(-1) & (s_all >= thresh)
        else:
            sel = valid.view(-1)
        flat = flat_all[sel]
        b, s, l = boxes.view(-1, 7)[flat], prob.reshape(-1)[flat], flat % nc
        return {"bbox3d": b, "scores": s, "labels": l}


class ROIBoxHead3D(nn.Module):
    """roi_heads/box_head_3d/box_head.py."""

    def __init__(self, cfg):
        super().__init__()
        self.feature_extractor = FPN2MLPFeatureExtractor(cfg)
        self.predictor = FPNPredictor(cfg)
        self.post_processor = PostProcessor(cfg)

        self.loss_evaluator = T.ROILoss(cfg)
        self.sep = SeperateClassifier(cfg.MODEL.SEPARATE_CLASSES_ID, len(cfg.INPUT.CLASSES))

    def _forward_grouped(self, roi_features, proposals, sep_id, targets):
        """seperate_subsample / roi_cross_entropy_seperated / roi_box_loss_seperated / post_processor
        (seperate_classifier.py:111-176,299-321)."""
        sep = self.sep
        if self.training:
            tg = sep.group_targets(targets)
            ps, ls, rs, ids = [], [], [], []
            for gi in range(sep.group_num):
                p, l, r = self.loss_evaluator.subsample(proposals[sep_id == gi], tg[gi]["bbox3d"], tg[gi]["labels"])
                ps.append(p); ls.append(l); rs.append(r)
                ids.append(torch.full((p.shape[0],), gi, dtype=torch.int64, device=p.device))
            proposals, labels, reg_targets, sep_id = torch.cat(ps), torch.cat(ls), torch.cat(rs), torch.cat(ids)
        x = self.feature_extractor(roi_features, proposals)
        logits, reg = self.predictor(x)
        assert logits.shape[1] == sep.total_classes
        n = logits.shape[0]
        out = {} if self.training else []
        for gi in range(sep.group_num):
            idx = torch.nonzero(sep_id == gi).view(-1)
            cols = torch.tensor(sep.grouped_classes[gi], device=logits.device)
            lg = logits[idx][:, cols]
            rg = reg.view(n, -1, 7)[:, cols, :].reshape(n, -1)[idx]
            if self.training:
                c, b = self.loss_evaluator(lg, rg, proposals[idx], labels[idx], reg_targets[idx])
                out[f"loss_classifier_roi_{gi}"], out[f"loss_box_reg_roi_{gi}"] = c, b
            else:
                res = self.post_processor(lg, rg.contiguous(), proposals[idx])
                res["labels"] = sep.org_label(gi, res["labels"])
                out.append(res)
        if self.training:
            return out
        return {k: torch.cat([r[k] for r in out]) for k in ("bbox3d", "scores", "labels")}

    def forward(self, roi_features, proposals, targets=None, sep_id=None):
        if sep_id is not None:
            return self._forward_grouped(roi_features, proposals, sep_id, targets)
        if self.training:                                                        # box_head.py:96-149
            proposals, labels, reg_targets = self.loss_evaluator.subsample(proposals, targets["bbox3d"],
                                                                           targets["labels"])
        x = self.feature_extractor(roi_features, proposals)
        logits, reg = self.predictor(x)
        if not self.training:
            return self.post_processor(logits, reg, proposals)
        cls_loss, box_loss = self.loss_evaluator(logits, reg, proposals, labels, reg_targets)
        return {"loss_classifier_roi": cls_loss, "loss_box_reg_roi": box_loss}


class _RoiHeads(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.box = ROIBoxHead3D(cfg)


class SparseRCNN(nn.Module):
    """Inference of one batch: points = [coords int64 [N,3|4], feats fp32 [N,9]]."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone = build_backbone(cfg)
        self.rpn = RPNModule(cfg)
        self.roi_heads = _RoiHeads(cfg)
        self.class_to_label = class_to_label(cfg.INPUT.CLASSES)

    def forward(self, points, targets=None, return_intermediates=False):
        """eval: detections dict.  train: dict of the four losses (targets = {"bbox3d" [M,7] yx_zb, "labels" [M]})."""
        if self.training:
            if targets is None:
                raise ValueError("In training mode, targets should be passed")
            rpn_features, roi_features = self.backbone(points)
            out = self.rpn(rpn_features, targets)
            proposals, rpn_losses = out[0].clone(), out[-1]
            sep_id = out[2] if len(out) == 4 else None
            proposals[:, 3:6] = torch.clamp(proposals[:, 3:6], min=0.001)
            losses = dict(self.roi_heads.box(roi_features, proposals, targets, sep_id=sep_id))
            losses.update(rpn_losses)
            return losses
        with torch.no_grad():
            return self._forward_eval(points, return_intermediates)

    def _forward_eval(self, points, return_intermediates=False):
        return self.stage_tail(self.backbone(points), return_intermediates)

    # the three stages of a pipelined inference pass (serving.BuildingPipeline); stage_tail(backbone(points)) is the
    # plain pass
    def stage_geometry(self, points):
        return self.backbone.stage_geometry(points)

    def stage_features(self, net):
        return self.backbone.stage_features(net)

    def stage_tail(self, features, return_intermediates=False):
        rpn_features, roi_features = features
        out = self.rpn(rpn_features)
        proposals, objectness = out[0].clone(), out[1]
        sep_id = out[2] if len(out) == 3 else None
        proposals[:, 3:6] = torch.clamp(proposals[:, 3:6], min=0.001)           # BoxList3D.clamp_size
        result = self.roi_heads.box(roi_features, proposals, sep_id=sep_id)
        if return_intermediates:
            return result, {"rpn_features": rpn_features, "roi_features": roi_features,
                            "proposals": proposals, "objectness": objectness}
        return result


def build_detection_model(cfg):
    return SparseRCNN(cfg)

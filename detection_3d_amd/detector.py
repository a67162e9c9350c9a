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
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from ._lib import check, host_word, lib, ptr, stream_of
from ._lib import ints as _ints
from . import sparseconvnet as scn
from . import training as T
from .config import class_to_label
from .timeline import mark
from .roi_align_rotated_3d import (roi_align_rotated_3d_sparse, roi_align_rotated_3d_sparse_into,
                                    roi_align_rotated_3d_sparse_levels_into, roi_prepare)
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


_ANCHORS_ONE_LAUNCH = os.environ.get("D3D_ANCHOR_LAUNCHES", "one") != "per-map"
_FUSED_RPN_HEAD = os.environ.get("D3D_RPN_HEAD", "fused") != "gemm"     # "gemm": the library-GEMM form, for A/B runs
# fc7 and the predictor behind fc6 as ONE launch (d3d_mlp_heads) instead of two activations and three library GEMMs: off by
# default -- a workgroup owns 32 rows and all 512 columns, so 1000 proposals occupy 32 CUs, whose matrix cores need 27 us
# for the product; the launch takes 47 us against 37 us for the five (D3D_BOX_MLP=fused switches it on)
_FUSED_BOX_MLP = os.environ.get("D3D_BOX_MLP", "gemm") == "fused"


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
        consts = getattr(self, "_anchor_consts", None)
        if consts is None:      # host copies of the (constant) cell anchors and strides as C arrays, made once
            from ._lib import floats
            consts = self._anchor_consts = [(floats([float(v) for row in b.tolist() for v in row]), b.shape[0],
                                             floats([float(v) for v in st.tolist()]))
                                            for b, st in zip(self.cell_anchors, self.strides)]
        stream = stream_of()
        maps = feature_maps_sparse
        if (_ANCHORS_ONE_LAUNCH and 1 <= len(maps) <= 6 and A <= 4 and all(c[1] == A for c in consts[:len(maps)])
                and len(consts) >= len(maps) and all(f.metadata is maps[0].metadata for f in maps)):
            packed = getattr(self, "_anchor_consts_maps", None)
            if packed is None or packed[0] != len(maps):
                from ._lib import floats
                packed = self._anchor_consts_maps = (
                    len(maps),
                    floats([float(v) for b in self.cell_anchors[:len(maps)] for row in b.tolist() for v in row]),
                    floats([float(v) for st in self.strides[:len(maps)].tolist() for v in st]))
            sizes = _ints(tuple(int(v) for f in maps for v in scn.SCN._size3(f.spatial_size)))
            check(lib().d3d_anchors_maps(maps[0].metadata._h, len(maps), sizes, packed[1], A, packed[2],
                                         float(self.voxel_scale), ptr(out), stream))
            return out
        for (base, n_base, stride), fmap, n in zip(consts, feature_maps_sparse, ns):
            if n:
                check(lib().d3d_anchors(fmap.metadata._h, _ints(scn.SCN._size3(fmap.spatial_size)), base, n_base,
                                        stride, float(self.voxel_scale), ptr(out[o * A:(o + n) * A]), stream))
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

    def _fused_ok(self, features):
        c = self.conv.weight.shape[0]
        return (_FUSED_RPN_HEAD and c in (128, 256) and 1 <= len(features) <= 6 and
                all(f.is_cuda and f.dtype == torch.float32 and f.is_contiguous() and f.dim() == 2 and f.shape[1] == c
                    for f in features))

    def _packed(self):
        """The operands of d3d_rpn_head: W[co][4g+j] -> [g][co][j]; objectness and regression weights stacked into one
        [8a, C] operand padded to whole 32-column tiles.  Re-made when a parameter changes (in-place updates bump
        `_version`, loads and `.to()` replace the storage)."""
        ps = (self.conv.weight, self.conv.bias, self.cls_logits.weight, self.cls_logits.bias,
              self.bbox_pred.weight, self.bbox_pred.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_pack", (None,))[0] != key:
            c = self.conv.weight.shape[0]

            def pack(w):                                    # [cout, c] -> [c/4, cout, 4]
                return w.view(w.shape[0], c // 4, 4).permute(1, 0, 2).contiguous()

            w2 = torch.cat([self.cls_logits.weight.detach().view(-1, c), self.bbox_pred.weight.detach().view(-1, c)], 0)
            pad = (-w2.shape[0]) % 32
            if pad:
                w2 = torch.cat([w2, w2.new_zeros(pad, c)], 0)
            self._pack = (key, pack(self.conv.weight.detach().view(c, c)), self.conv.bias.detach().contiguous(),
                          pack(w2), torch.cat([self.cls_logits.bias.detach(), self.bbox_pred.bias.detach()]).contiguous())
        return self._pack[1:]

    def _forward_fused(self, features):
        """One d3d_rpn_head launch over the maps' rows where they lie (no concatenation, no library GEMMs)."""
        import ctypes
        w1, b1, w2, b2 = self._packed()
        a = self.num_anchors_per_location * self.seperate_rpn
        n = sum(f.shape[0] for f in features)
        dev = features[0].device
        obj = torch.empty((n, a), dtype=torch.float32, device=dev)
        reg = torch.empty((n, 7 * a), dtype=torch.float32, device=dev)
        maps = (ctypes.c_void_p * len(features))(*[f.data_ptr() for f in features])
        check(lib().d3d_rpn_head(maps, _ints(tuple(int(f.shape[0]) for f in features)), len(features),
                                 int(features[0].shape[1]), ptr(w1), ptr(b1), ptr(w2), ptr(b2), a, ptr(obj), ptr(reg),
                                 stream_of()))
        return obj.view(-1, self.seperate_rpn), reg.view(-1, 7 * self.seperate_rpn)

    def forward(self, features):
        """features: list of [n_s, C]  ->  objectness [sum n_s*A], regression [sum n_s*A, 7] in the
        flattening order of cat_scales_obj_reg (:19-77): scale, site, anchor."""
        if not torch.is_grad_enabled() and self._fused_ok(features):
            return self._forward_fused(features)
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


class PaddedProposals(object):
    """The RPN's proposals of one example before the survivor count of its NMS is read back: `boxes` [P, 7] and `scores`
    [P] padded to P = FPN_POST_NMS_TOP_N rows, `count` int32 [1] on the device = the real rows (the first ones)."""

    def __init__(self, boxes, scores, count, readback=None):
        # readback = (pinned word d3d_gather_kept stored the count to, event recorded behind that launch): the host waits
        # for the event alone, not for the pooler launches that follow it on the stream
        self.boxes, self.scores, self.count, self._readback = boxes, scores, count, readback

    def resolve(self):
        """the one host synchronisation -> (boxes [n, 7], scores [n])"""
        if self._readback is not None:
            word, copied = self._readback
            copied.synchronize()
            n = int(word[0])
        else:
            n = int(self.count.item())
        return self.boxes[:n], self.scores[:n]


_DEFER_PROPOSALS = os.environ.get("D3D_DEFER_PROPOSALS", "1") != "0"


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
    def select_proposals(self, objectness, box_regression, anchors, train, defer=False):
        """defer (inference): -> PaddedProposals -- the survivors gathered into `post` rows with their count left on the
        device, so that the pooler is enqueued behind the NMS without waiting for its read-back.
        sigmoid -> top-k -> gathers -> decode (inference_3d.py:105-123) are ONE launch (d3d_topk_segments; equal scores:
        lower anchor index first, the order the oracle port defines)."""
        pre, post = self.top_n[bool(train)]
        logits = objectness.reshape(-1).contiguous()
        k = min(pre, logits.shape[0])
        reg = box_regression.contiguous()
        if k == 0 or k > box_ops.topk_max() or reg.shape[1] != 7:
            scores = logits.sigmoid()
            order = torch.sort(scores, descending=True, stable=True)[1][:k]    # the same order, in tensor ops
            scores_k = scores[order]
            proposals = box_ops.box_decode(reg[order], anchors[order])
        else:
            tk = box_ops.topk_segments(logits, k, sigmoid=True, reg=reg, anchors=anchors.contiguous(), want_idx64=False)
            proposals, scores_k = tk["props"][0, :k], tk["scores"][0, :k]
        if defer and 0 < k <= 2000 and post > 100:
            keep, nk = box_ops.nms_3d_batched(proposals, None, None, k, self.nms_thresh, self.nms_aug_thickness, post)
            # the survivors, padded to `post` rows, sizes clamped (BoxList3D.clamp_size): one launch, which also
            # stores the count to a pinned word for the host
            word, stored = host_word(proposals.device, "rpn")
            boxes, scores_p = box_ops.gather_kept(proposals, scores_k, keep, nk, min(post, k), 0.001, count_host=word)
            stored.record()
            return PaddedProposals(boxes, scores_p, nk, (word, stored))
        keep = box_ops.nms_3d_presorted(proposals, self.nms_thresh, self.nms_aug_thickness, max_proposals=post,
                                        flag='rpn_post')                        # scores_k is sorted: no re-sort
        return proposals[keep], scores_k[keep]

    @torch.no_grad()
    def select_proposals_segments(self, objectness, box_regression, anchors, example, n_examples, train):
        """RPNPostProcessor.forward_for_single_feature_map (rpn/inference_3d.py:82-163) for every (example, class group)
        pair at once: objectness [n, G], box_regression [n, 7 G], anchors [n, 7], example int64 [n] (None: one example).
        Segment s = example * G + group gets its own top-k over the example's anchors; ONE launch selects, gathers and
        decodes for all segments (d3d_topk_segments), one batched NMS follows, ONE read-back returns the survivor counts.
        -> list over segments of (proposals [m_s, 7], scores [m_s])."""
        pre, post = self.top_n[bool(train)]
        n, G = objectness.shape
        B = int(n_examples)
        S = B * G
        k = min(pre, n)
        if k == 0:
            z = anchors.new_zeros
            return [(z((0, 7)), z((0,))) for _ in range(S)]
        assert k <= box_ops.topk_max(), k
        ex32 = example.to(torch.int32).contiguous() if B > 1 else None
        tk = box_ops.topk_segments(objectness.contiguous(), k, n=n, elem_stride=G, group_stride=1, n_groups=G, example=ex32,
                                   n_examples=B, sigmoid=True, reg=box_regression.contiguous(),
                                   anchors=anchors.contiguous(), want_idx64=False)
        props, sk = tk["props"].view(-1, 7), tk["scores"].reshape(-1)             # [S k, 7]: inference_3d.py:109-123
        keep, nk = box_ops.nms_3d_batched(props, None, tk["counts"], k, self.nms_thresh, self.nms_aug_thickness, post,
                                          segments=S)
        out = []
        for s_, m in enumerate(nk.tolist()):                                # the one host synchronisation
            sel = keep[s_, :m].long()
            out.append((props[sel], sk[sel]))
        return out

    def forward(self, features_sparse, targets=None, n_examples=1, defer=False):
        """eval: (proposals, objectness) -- or, with defer, a PaddedProposals when the path allows it.
        train: (proposals incl. GT boxes, objectness, loss dict)
        (rpn_sparse3d.py:233-270, rpn/inference_3d.py:53-80,180-199).  n_examples > 1 (inference): returns
        (proposals, objectness, sep_id or None, example_id) with the rows ordered by example (then class group)."""
        objectness, box_regression = self.head([f.features for f in features_sparse])
        with torch.no_grad():
            anchors = self.anchor_generator.forward_cat(features_sparse)
        mark("rpn head + anchors")
        assert objectness.shape[0] == box_regression.shape[0] == anchors.shape[0]
        grouped = self.sep.need_seperate and self.head.seperate_rpn > 1
        if n_examples > 1:
            if self.training:
                raise NotImplementedError("training with more than one example per batch is not built "
                                          "(the reference's configs train with IMS_PER_BATCH 1)")
            A = self.anchor_generator.num_anchors_per_location()
            example = torch.cat([f.get_spatial_locations()[:, 3].repeat_interleave(A) for f in features_sparse])
            segs = self.select_proposals_segments(objectness.detach(), box_regression.detach(), anchors, example,
                                                  n_examples, False)
            G = objectness.shape[1]
            dev = anchors.device
            sep_id = torch.cat([torch.full((p.shape[0],), i % G, dtype=torch.int64, device=dev) for i, (p, _) in enumerate(segs)])
            ex_id = torch.cat([torch.full((p.shape[0],), i // G, dtype=torch.int64, device=dev) for i, (p, _) in enumerate(segs)])
            return (torch.cat([p for p, _ in segs]), torch.cat([sc for _, sc in segs]), sep_id if grouped else None, ex_id)
        if grouped:
            return self._forward_grouped(anchors, objectness, box_regression, targets)
        out = self.select_proposals(objectness.detach(), box_regression.detach(), anchors, self.training,
                                    defer=defer and not self.training and _DEFER_PROPOSALS)
        if not self.training:
            return out
        proposals, scores = out
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
        # the groups' top-k / decode / NMS as segments of ONE launch set (seperate_rpn_selector loops the selector)
        segs = self.select_proposals_segments(objectness.detach(), box_regression.detach(), anchors, None, 1,
                                              self.training)
        for gi in range(self.sep.group_num):
            obj_g, reg_g = objectness[:, gi], box_regression[:, 7 * gi:7 * gi + 7]
            p, sc = segs[gi]
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


_ROI_ONE_LAUNCH = os.environ.get("D3D_ROI_LAUNCHES", "one") != "per-level"   # "per-level": one launch per map (A/B)


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

    def pool_metric(self, x, boxes_metric, voxel_scale, channels_inner=True, batch_ids=None, count=None):
        """Inference: metric proposals -> pooled features with ONE pre-processing launch (pixels, RoI format and FPN
        level: d3d_roi_prepare, bit-identical to convert_to_roi_format / map_levels) and one launch per level that
        fills its RoIs' slots of the result in place (no nonzero / index_put, no host synchronisation)."""
        ph, pw, pz = self.output_size
        rois, levels = roi_prepare(boxes_metric, voxel_scale, self.scales, self.canonical_size, batch_ids, count)
        K, C = rois.shape[0], x[0].features.shape[1]
        out = torch.empty((K, ph, pw, C, pz) if channels_inner else (K, C, ph, pw, pz), dtype=torch.float32,
                          device=rois.device)
        if _ROI_ONE_LAUNCH and len(x) <= 4 and all(f.metadata is x[0].metadata for f in x):
            return roi_align_rotated_3d_sparse_levels_into(out, x, rois, self.scales, self.sampling_ratio, levels,
                                                           channels_inner=channels_inner)
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


def _pack_linear(w):
    """nn.Linear weight [cout, c] -> the [c/4, cout, 4] operand of d3d_rpn_head / d3d_mlp_heads (rows padded to 32)"""
    cout, c = w.shape
    pad = (-cout) % 32
    if pad:
        w = torch.cat([w, w.new_zeros(pad, c)], 0)
    return w.view(w.shape[0], c // 4, 4).permute(1, 0, 2).contiguous()


def _mlp_rows_ok(x):
    return (_FUSED_BOX_MLP and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.is_contiguous() and x.shape[1] in (128, 256, 512) and x.shape[0] > 0)


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

    def rows_path_ok(self, padded=False):
        """whether `_forward_rows` serves this configuration (padded: with the proposal count still on the device, which
        needs the level map of a multi-level pooler to switch the padding rows off)"""
        conv, bn = self.conv3d[0], self.conv3d[1]
        return (not torch.is_grad_enabled() and tuple(conv.kernel_size) == (1, 1, self.pooler.output_size[2])
                and tuple(conv.stride) == (1, 1, 1) and (bn.training or not bn.track_running_stats)
                and (not padded or len(self.pooler.scales) > 1))

    def forward_padded(self, x0, padded):
        """PaddedProposals (sizes clamped by d3d_gather_kept) -> (box features [n, rep], proposals [n, 7], scores [n]): the pooler is
        enqueued over all padded rows, then the count is read back and the head runs over the real ones."""
        pooled = self.pooler.pool_metric(x0, padded.boxes, self.voxel_scale, channels_inner=True, count=padded.count)
        proposals, scores = padded.resolve()
        n = proposals.shape[0]
        if n == 0:
            return self.forward(x0, proposals), proposals, scores
        return self._forward_rows(x0, proposals, pooled=pooled[:n]), proposals, scores

    def _forward_rows(self, x0, p, batch_ids=None, pooled=None):
        """Inference path without layout changes (p: metric proposals): the pooler writes [K, ph, pw, C, pz], whose rows feed the
        [1,1,pz] convolution as a GEMM; BatchNorm3d + ReLU is one row-wise BatchNorm over [K*ph*pw, rep]
        (batch statistics, biased variance: F.batch_norm in training mode), fc6 reads the rows in place."""
        conv, bn = self.conv3d[0], self.conv3d[1]
        if pooled is None:
            pooled = self.pooler.pool_metric(x0, p, self.voxel_scale, channels_inner=True, batch_ids=batch_ids)
        mark("rois pooled")
        K, ph, pw, C, pz = pooled.shape
        y = torch.addmm(conv.bias, pooled.view(K * ph * pw, C * pz), conv.weight.view(conv.out_channels, C * pz).t())
        rep = y.shape[1]
        out, sm, si = y.new_empty(0), y.new_empty(rep), y.new_empty(rep)
        run = getattr(self, "_bn_running", None)      # throw-away running statistics (momentum 0: overwritten per call)
        if run is None or run[0].device != y.device or run[0].shape[0] != rep:
            run = self._bn_running = (y.new_zeros(rep), y.new_ones(rep))
        SCN.BatchNormalization_updateOutput(y, out, sm, si, run[0], run[1], bn.weight, bn.bias, bn.eps, 0.0, True, 0.0)
        h = torch.addmm(self.fc6.bias, out.view(K, ph * pw * rep), self._fc6_rows_weight(ph * pw).t())
        h = self._fc7_and_heads(h)
        mark("box features")
        return h

    def _fc7_packed(self):
        w, b = self.fc7.weight, self.fc7.bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if getattr(self, "_fc7_pack", (None,))[0] != key:
            self._fc7_pack = (key, _pack_linear(w.detach()), b.detach().contiguous())
        return self._fc7_pack[1:]

    def _fc7_and_heads(self, h6):
        """relu(fc7(relu(h6))) -- and, when the box head has handed its predictor over (`_heads`), cls_score and bbox_pred
        of the result in the same launch (d3d_mlp_heads; they ride on the returned tensor for FPNPredictor.forward, which
        produces the same bits when it is called on its own)."""
        if not (_mlp_rows_ok(h6) and self.fc7.in_features == self.fc7.out_features == h6.shape[1]):
            return F.relu(self.fc7(F.relu(h6)))
        w1, b1 = self._fc7_packed()
        x = torch.empty_like(h6)
        pred = getattr(self, "_heads", None)
        if pred is not None and pred.fusable(x):
            w2, b2, a, key = pred.packed()
            logits = torch.empty((x.shape[0], a), dtype=torch.float32, device=x.device)
            reg = torch.empty((x.shape[0], 7 * a), dtype=torch.float32, device=x.device)
            check(lib().d3d_mlp_heads(ptr(h6), x.shape[0], x.shape[1], 1, ptr(w1), ptr(b1), ptr(x), ptr(w2), ptr(b2), a,
                                      ptr(logits), ptr(reg), stream_of()))
            x._d3d_heads = (key, logits, reg)
        else:
            check(lib().d3d_mlp_heads(ptr(h6), x.shape[0], x.shape[1], 1, ptr(w1), ptr(b1), ptr(x), None, None, 0, None,
                                      None, stream_of()))
        return x

    def forward(self, x0, proposals, batch_ids=None):
        """batch_ids: int32 [K] example of every proposal (inference with several examples per batch; the RoI op reads
        the example's sites, poolers_3d.py:112-118); None = one example."""
        if self.rows_path_ok() and proposals.shape[0] > 0:
            return self._forward_rows(x0, proposals, batch_ids)                 # metric boxes: pixels on the device
        if batch_ids is not None:
            raise NotImplementedError("several examples per batch: only the inference path of the box head is built")
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

    def packed(self):
        """-> (cls_score and bbox_pred stacked as one packed operand, their biases, classes, version key)"""
        ps = (self.cls_score.weight, self.cls_score.bias, self.bbox_pred.weight, self.bbox_pred.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_pack", (None,))[0] != key:
            w2 = torch.cat([self.cls_score.weight.detach(), self.bbox_pred.weight.detach()], 0)
            self._pack = (key, _pack_linear(w2), torch.cat([self.cls_score.bias.detach(), self.bbox_pred.bias.detach()]).contiguous())
        return self._pack[1], self._pack[2], self.cls_score.out_features, key

    def fusable(self, x):
        return (_mlp_rows_ok(x) and self.cls_score.in_features == x.shape[1]
                and self.bbox_pred.out_features == 7 * self.cls_score.out_features)

    def forward(self, x):
        if self.fusable(x):
            w2, b2, a, key = self.packed()
            done = getattr(x, "_d3d_heads", None)        # computed with fc7 in one launch (FPN2MLPFeatureExtractor)
            if done is not None and done[0] == key:
                return done[1], done[2]
            logits = torch.empty((x.shape[0], a), dtype=torch.float32, device=x.device)
            reg = torch.empty((x.shape[0], 7 * a), dtype=torch.float32, device=x.device)
            check(lib().d3d_mlp_heads(ptr(x), x.shape[0], x.shape[1], 0, None, None, None, ptr(w2), ptr(b2), a, ptr(logits),
                                      ptr(reg), stream_of()))
            return logits, reg
        return self.cls_score(x), self.bbox_pred(x)


_POST_SELECT_CAP = []


def _POST_SELECT_MAX():
    if not _POST_SELECT_CAP:
        _POST_SELECT_CAP.append(int(lib().d3d_post_select_max()))
    return _POST_SELECT_CAP[0]


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
        n_max = min(K, 2000)                                                     # pre_max_size of rotate_nms_3d
        if n_max <= box_ops.topk_max():
            # class j's list = the RoIs with prob[:, j] > thresh, best first (equal scores: lower RoI first), as box
            # indices roi * nc + j: one launch (d3d_topk_segments over the nc - 1 foreground columns)
            tk = box_ops.topk_segments(prob.view(-1)[1:], n_max, n=K, elem_stride=nc, group_stride=1, n_groups=nc - 1,
                                       min_value=self.score_thresh, idx_map=(nc, 1, 1), want_idx64=False, want_idx32=True)
            order, counts = tk["idx32"], tk["counts"]
        else:
            sc = torch.empty((nc - 1, K), dtype=torch.float32, device=dev)
            counts = torch.empty((nc - 1,), dtype=torch.int32, device=dev)
            check(lib().d3d_post_scores(ptr(prob), K, nc, float(self.score_thresh), ptr(sc), ptr(counts), stream_of()))
            idx = torch.sort(sc, dim=1, descending=True, stable=True)[1]
            order = torch.empty((nc - 1, K), dtype=torch.int32, device=dev)
            check(lib().d3d_post_order(ptr(idx), K, nc, ptr(order), stream_of()))
        keep, nk = box_ops.nms_3d_batched(boxes.view(-1, 7), order, counts, n_max, self.nms, self.nms_aug_thickness,
                                          500)                                   # post_max_size (boxlist_ops_3d.py)
        N = (nc - 1) * n_max
        if N <= _POST_SELECT_MAX():
            # the cut to detections_per_img and the final gathers in one launch (d3d_post_select), one read-back
            out_b = torch.empty((N, 7), dtype=torch.float32, device=dev)
            out_s = torch.empty((N,), dtype=torch.float32, device=dev)
            out_l = torch.empty((N,), dtype=torch.int64, device=dev)
            out_n, stored = host_word(dev, "detections")     # pinned: the kernel's store is the read-back
            boxes = boxes.contiguous()
            check(lib().d3d_post_select(ptr(keep), ptr(nk), nc - 1, n_max, ptr(prob), ptr(boxes), nc,
                                        int(self.detections_per_img), ptr(out_b), ptr(out_s), ptr(out_l), ptr(out_n),
                                        stream_of()))
            stored.record()
            stored.synchronize()                                                 # the one host synchronisation
            n = int(out_n[0])
            return {"bbox3d": out_b[:n], "scores": out_s[:n], "labels": out_l[:n]}
        s_all = torch.empty((N,), dtype=torch.float32, device=dev)
        flat_all = torch.empty((N,), dtype=torch.int64, device=dev)
        check(lib().d3d_post_gather(ptr(keep), ptr(nk), nc - 1, n_max, ptr(prob), ptr(s_all), ptr(flat_all), stream_of()))
        if 0 < self.detections_per_img < s_all.shape[0]:                         # :140-148 without a second read-back:
            # the D-th largest score; with fewer than D survivors it is a padding entry (-1) and every survivor stays
            thresh = torch.sort(s_all, descending=True)[0][self.detections_per_img - 1]
            sel = s_all >= thresh.clamp_min(0.0)                                 # survivors have scores > 0, padding -1
        else:
            sel = s_all >= 0.0
        flat = flat_all[sel]                                                     # the one host synchronisation
        b, s, l = boxes.view(-1, 7)[flat], prob.reshape(-1)[flat], flat % nc
        return {"bbox3d": b, "scores": s, "labels": l}

    @torch.no_grad()
    def forward_segments(self, class_logits, box_regression, proposals, seg_id, seg_sizes, seg_group, grouped_classes):
        """The post-processing of several (example, class group) row segments in ONE launch set: what
        SeperateClassifier.post_processor (seperate_classifier.py:299-321) and the per-image loop of
        box_head_3d/inference.py:66-99 do one call at a time.
          class_logits [K, T], box_regression [K, 7 T]: all rows, all T = total class columns;
          seg_id int64 [K]: segment of every row; seg_sizes: rows per segment (host list, for the NMS width);
          seg_group[s]: class group of segment s; grouped_classes[g]: the group's columns, background first.
        Every (segment, foreground class of its group) is one NMS segment; the top DETECTIONS_PER_IMG are taken per
        segment.  -> list over segments of {"bbox3d", "scores", "labels" (inside the group)}."""
        K, T = class_logits.shape
        S, G = len(seg_sizes), len(grouped_classes)
        dev = class_logits.device
        empty = {"bbox3d": class_logits.new_zeros((0, 7)), "scores": class_logits.new_zeros((0,)),
                 "labels": torch.zeros(0, dtype=torch.int64, device=dev)}
        cmax = max(len(g) for g in grouped_classes)
        if K == 0 or cmax < 2:
            return [dict(empty) for _ in range(S)]
        cols = torch.tensor([list(g) + [g[0]] * (cmax - len(g)) for g in grouped_classes], dtype=torch.int64, device=dev)
        group_of_row = torch.tensor(list(seg_group), dtype=torch.int64, device=dev)[seg_id]              # [K]
        # softmax over each group's own columns (the same row-wise op as the per-group call), then per row the
        # probabilities of its group; padded class slots get probability 0
        prob = class_logits.new_zeros((K, cmax))
        for g, classes in enumerate(grouped_classes):
            pg = F.softmax(class_logits[:, cols[g, :len(classes)]], -1)
            prob[:, :len(classes)] = torch.where((group_of_row == g).view(-1, 1), pg, prob[:, :len(classes)])
        c = cols[group_of_row]                                                                            # [K, cmax]
        reg = box_regression.view(K, T, 7).gather(1, c[:, :, None].expand(-1, -1, 7)).reshape(K, cmax * 7)
        boxes = box_ops.box_decode(reg, proposals, self.weights)                                          # [K, 7 cmax]
        # NMS segments (s, j): rows of segment s with prob[:, j] > thresh, descending score, ties: lower row first
        nseg = S * (cmax - 1)
        member = seg_id.view(1, -1) == torch.arange(S, device=dev).view(-1, 1)                            # [S, K]
        p = prob[:, 1:].t()                                                                               # [cmax-1, K]
        cand = (p > self.score_thresh)[None] & member[:, None, :]                                         # [S, cmax-1, K]
        sc = torch.where(cand, p[None], p.new_full((), -1.0)).reshape(nseg, K).contiguous()
        n_max = max(1, min(max(seg_sizes), 2000))                                                         # pre_max_size
        assert n_max <= box_ops.topk_max()
        # every (segment, class) row's candidates, best first (equal scores: lower row first): one launch
        tk = box_ops.topk_segments(sc, n_max, n=K, elem_stride=1, group_stride=K, n_groups=nseg, min_value=self.score_thresh,
                                   want_idx64=False, want_idx32=True)
        slot = (torch.arange(cmax - 1, device=dev, dtype=torch.int32) + 1).repeat(S).view(nseg, 1)
        order = (tk["idx32"] * cmax + slot).contiguous()
        counts = tk["counts"]
        keep, nk = box_ops.nms_3d_batched(boxes.view(-1, 7), order, counts, n_max, self.nms, self.nms_aug_thickness, 500)
        valid = torch.arange(n_max, device=dev).view(1, -1) < nk.view(-1, 1)                              # [nseg, n_max]
        flat = torch.where(valid, keep, torch.zeros_like(keep)).long()                                    # row * cmax + slot
        s_all = torch.where(valid, prob.reshape(-1)[flat], prob.new_full((), -1.0)).view(S, -1)
        valid, flat = valid.view(S, -1), flat.view(S, -1)
        if 0 < self.detections_per_img < s_all.shape[1]:                                                 # :140-148 per segment
            if self.detections_per_img <= box_ops.topk_max():
                top = box_ops.topk_segments(s_all.contiguous(), self.detections_per_img, n=s_all.shape[1], elem_stride=1,
                                            group_stride=s_all.shape[1], n_groups=S, want_idx64=False)["scores"]
                thresh = top[:, self.detections_per_img - 1]
            else:
                thresh = torch.sort(s_all, dim=1, descending=True)[0][:, self.detections_per_img - 1]
            sel = valid & (s_all >= thresh.view(-1, 1))
        else:
            sel = valid
        n_sel = sel.sum(1).tolist()                                                                       # the one synchronisation
        picked = flat[sel]                                                                                # segment-major
        b, sc_, lab = boxes.view(-1, 7)[picked], prob.reshape(-1)[picked], picked % cmax
        out, o = [], 0
        for m in n_sel:
            out.append({"bbox3d": b[o:o + m], "scores": sc_[o:o + m], "labels": lab[o:o + m]})
            o += m
        return out

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
            thresh = torch.sort(s_all, descending=True)[0][self.detections_per_img - 1]
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
        # fc7 and the predictor share a launch at inference; a plain attribute, not a second registration of the module
        object.__setattr__(self.feature_extractor, "_heads", self.predictor)
        self.post_processor = PostProcessor(cfg)

        self.loss_evaluator = T.ROILoss(cfg)
        self.sep = SeperateClassifier(cfg.MODEL.SEPARATE_CLASSES_ID, len(cfg.INPUT.CLASSES))

    def _forward_grouped(self, roi_features, proposals, sep_id, targets):
        """seperate_subsample / roi_cross_entropy_seperated / roi_box_loss_seperated / post_processor
        (seperate_classifier.py:111-176,299-321)."""
        sep = self.sep
        if not self.training:
            return self._forward_eval_segments(roi_features, proposals, sep_id, None, 1)[0]
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
        ids_g = [torch.nonzero(sep_id == gi).view(-1) for gi in range(sep.group_num)]
        out = {}
        for gi, (lg, rg) in enumerate(zip(sep.seperate_pred_logits(logits, ids_g), sep.seperate_pred_box(reg, ids_g))):
            idx = ids_g[gi]
            c, b = self.loss_evaluator(lg, rg, proposals[idx], labels[idx], reg_targets[idx])
            out[f"loss_classifier_roi_{gi}"], out[f"loss_box_reg_roi_{gi}"] = c, b
        return out

    def _forward_eval_segments(self, roi_features, proposals, sep_id, example_id, n_examples):
        """Inference for rows ordered by (example, class group): ONE box-head pass over all RoIs (as the reference:
        the head's BatchNorm3d sees every RoI of the batch), one post-processing launch set for all segments.
        -> list over examples of the detections dict (labels are the original class ids)."""
        sep = self.sep
        G = sep.group_num if (sep.need_seperate and sep_id is not None) else 1
        dev = proposals.device
        gid = sep_id if G > 1 else torch.zeros(proposals.shape[0], dtype=torch.int64, device=dev)
        ex = example_id if example_id is not None else torch.zeros_like(gid)
        seg_id = ex * G + gid
        S = n_examples * G
        seg_sizes = torch.bincount(seg_id, minlength=S).tolist()                  # host sizes of the segments
        batch_ids = ex.to(torch.int32).contiguous() if n_examples > 1 else None
        x = self.feature_extractor(roi_features, proposals, batch_ids)
        logits, reg = self.predictor(x)
        if G > 1:
            assert logits.shape[1] == sep.total_classes
            grouped = sep.grouped_classes
        else:
            grouped = [list(range(logits.shape[1]))]
        segs = self.post_processor.forward_segments(logits, reg, proposals, seg_id, seg_sizes,
                                                    [s_ % G for s_ in range(S)], grouped)
        results = []
        for b in range(n_examples):
            parts = []
            for g in range(G):
                r = segs[b * G + g]
                if G > 1:
                    r = dict(r, labels=sep.org_label(g, r["labels"]))
                parts.append(r)
            results.append({k: torch.cat([r[k] for r in parts]) for k in ("bbox3d", "scores", "labels")})
        return results

    def forward_padded(self, roi_features, padded):
        """inference from a PaddedProposals -> (detections dict, proposals [n, 7], scores [n])"""
        x, proposals, scores = self.feature_extractor.forward_padded(roi_features, padded)
        logits, reg = self.predictor(x)
        return self.post_processor(logits, reg, proposals), proposals, scores

    def forward(self, roi_features, proposals, targets=None, sep_id=None, example_id=None, n_examples=1):
        if n_examples > 1:
            assert not self.training
            return self._forward_eval_segments(roi_features, proposals, sep_id, example_id, n_examples)
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
            if self.batch_size_of(points) > 1:
                raise NotImplementedError("training with more than one example per batch is not built")
            rpn_features, roi_features = self.backbone(points[:2])
            out = self.rpn(rpn_features, targets)
            proposals, rpn_losses = out[0].clone(), out[-1]
            sep_id = out[2] if len(out) == 4 else None
            proposals[:, 3:6] = torch.clamp(proposals[:, 3:6], min=0.001)
            losses = dict(self.roi_heads.box(roi_features, proposals, targets, sep_id=sep_id))
            losses.update(rpn_losses)
            return losses
        with torch.no_grad():
            return self._forward_eval(points, return_intermediates)

    @staticmethod
    def batch_size_of(points):
        """examples in `points` = [coords int64 [N, 3|4], feats(, batch_size)]: the explicit third entry, else 1 + the
        largest batch index of the coordinates (one read-back; coordinates of 3 columns are one example)."""
        if len(points) > 2 and points[2]:
            return int(points[2])
        coords = points[0]
        if coords.shape[1] < 4 or coords.shape[0] == 0:
            return 1
        return int(coords[:, 3].max().item()) + 1

    def _forward_eval(self, points, return_intermediates=False):
        n_examples = self.batch_size_of(points)
        return self.stage_tail(self.backbone(points[:2]), return_intermediates, n_examples)

    # the three stages of a pipelined inference pass (serving.BuildingPipeline); stage_tail(backbone(points)) is the
    # plain pass
    def stage_geometry(self, points):
        return self.backbone.stage_geometry(points)

    def stage_features(self, net):
        return self.backbone.stage_features(net)

    def stage_tail(self, features, return_intermediates=False, n_examples=1):
        """-> detections dict of the example; for n_examples > 1 (coordinates with a batch column, examples listed one
        after the other as the reference's collate does) a list of such dicts, one per example."""
        rpn_features, roi_features = features
        mark("backbone done")
        out = self.rpn(rpn_features, n_examples=n_examples,
                       defer=self.roi_heads.box.feature_extractor.rows_path_ok(padded=True))
        mark("proposals")
        if isinstance(out, PaddedProposals):
            # the pooler goes out behind the NMS; the survivor count is read back while it runs
            result, proposals, objectness = self.roi_heads.box.forward_padded(roi_features, out)
            mark("detections")
            if return_intermediates:
                return result, {"rpn_features": rpn_features, "roi_features": roi_features,
                                "proposals": proposals, "objectness": objectness, "example_id": None,
                                "sep_id": None}
            return result
        proposals, objectness = out[0].clone(), out[1]
        example_id = None
        if n_examples > 1:
            sep_id, example_id = out[2], out[3]
        else:
            sep_id = out[2] if len(out) == 3 else None
        proposals[:, 3:6] = torch.clamp(proposals[:, 3:6], min=0.001)           # BoxList3D.clamp_size
        result = self.roi_heads.box(roi_features, proposals, sep_id=sep_id, example_id=example_id, n_examples=n_examples)
        mark("detections")
        if return_intermediates:
            return result, {"rpn_features": rpn_features, "roi_features": roi_features,
                            "proposals": proposals, "objectness": objectness, "example_id": example_id,
                            "sep_id": sep_id}
        return result


def build_detection_model(cfg):
    from . import tuned_gemm
    tuned_gemm.enable()         # library GEMMs of the tail: solutions picked ahead of time (no tuning at run time)
    return SparseRCNN(cfg)

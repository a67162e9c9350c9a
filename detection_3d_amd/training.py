"""Training rows: label generation and losses of the reference in tensor form,
optimizer / LR schedule surface, and the data-parallel wrapper.

    Matcher                            maskrcnn_benchmark/modeling/matcher.py:13-177
    BalancedPositiveNegativeSampler    modeling/balanced_positive_negative_sampler.py:5-68
    box_encode                         modeling/box_coder_3d.py:31-36 + second/pytorch/core/box_torch_ops.py:15-46
    smooth_l1_loss                     maskrcnn_benchmark/layers/smooth_l1_loss.py:15-49
    RPNLoss                            modeling/rpn/loss_3d.py:69-250
    ROILoss                            modeling/roi_heads/box_head_3d/loss.py:20-236
    make_optimizer / WarmupMultiStepLR maskrcnn_benchmark/solver/{build,lr_scheduler}.py
The IoU matrices come from the HIP kernel (box_ops.boxes_iou_3d); the rest is small tensor algebra.
"""
import math
from bisect import bisect_right

import torch
import torch.nn.functional as F

from . import box_ops


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False, yaw_threshold=3.1416 * 0.4):
        assert low_threshold <= high_threshold
        self.high_threshold, self.low_threshold = high_threshold, low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches
        self.yaw_threshold = yaw_threshold

    def __call__(self, match_quality_matrix, yaw_diff=None):
        """[M gt, N predictions] -> int64 [N]: matched gt index, -1 below low threshold, -2 between."""
        q = match_quality_matrix
        if q.numel() == 0:
            raise ValueError("No ground-truth or proposal boxes available for one of the images during training")
        if yaw_diff is not None and not self.yaw_threshold > 1.58:                 # matcher.py:50-55
            q = q * (torch.abs(yaw_diff) < self.yaw_threshold).float()
        matched_vals, matches = q.max(dim=0)
        all_matches = matches.clone()
        below = matched_vals < self.low_threshold
        between = (matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)
        matches[below] = Matcher.BELOW_LOW_THRESHOLD
        matches[between] = Matcher.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            # matcher.py:105-177: every gt keeps its best prediction(s) (ties included); predictions within
            # 0.05 of a gt's best (and above 0.02) that would be negatives are ignored instead
            highest = q.max(dim=1)[0]
            pairs = torch.nonzero(q == highest[:, None])
            upd = pairs[:, 1]
            matches[upd] = all_matches[upd]
            ignore_thr = torch.max(torch.full_like(highest, 0.02), highest - 0.05)
            ignore = (q > ignore_thr.view(-1, 1)).any(dim=0) & (matches == Matcher.BELOW_LOW_THRESHOLD)
            matches[ignore] = Matcher.BETWEEN_THRESHOLDS
        return matches


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction

    def __call__(self, labels):
        """labels: -1 ignored, 0 negative, >0 positive -> (pos index tensor, neg index tensor)."""
        positive = torch.nonzero(labels >= 1).squeeze(1)
        negative = torch.nonzero(labels == 0).squeeze(1)
        num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
        num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
        perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
        perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
        return positive[perm1], negative[perm2]


def box_encode(targets, anchors, weights=(1.0,) * 7):
    """BoxCoder3D.encode with smooth_dim=True; yaw residual wrapped to [-pi/2, pi/2)."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, wg, lg, hg, rg = torch.split(targets, 1, dim=-1)
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    enc = torch.cat([(xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / ha, wg / wa - 1, lg / la - 1,
                     hg / ha - 1, rg - ra], dim=-1)
    enc[:, -1] = box_ops.limit_period(enc[:, -1], 0.5, math.pi)
    return enc * torch.tensor(weights, dtype=enc.dtype, device=enc.device).view(1, 7)


def smooth_l1_loss(input, target, anchor_boxes, beta=1. / 9, size_average=True, yaw_loss_mode='Diff'):
    assert input.shape == target.shape and input.shape[1] == 7
    dif = torch.abs(input - target)
    mode = yaw_loss_mode.split('_')
    if mode[0] == 'SinDiff':
        dif_yaw = dif[:, -1]
        pred_yaw = input[:, -1] + anchor_boxes[:, -1]
        yaw = torch.where(torch.abs(pred_yaw) <= math.pi / 2, torch.sin(dif_yaw), dif_yaw)
        dif = torch.cat([dif[:, :-1], (yaw * (float(mode[1]) if len(mode) == 2 else 1.0)).unsqueeze(1)], 1)
    else:
        assert mode[0] == 'Diff'
    loss = torch.where(dif < beta, 0.5 * dif ** 2 / beta, dif - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()


class RPNLoss(object):
    def __init__(self, cfg):
        rpn = cfg.MODEL.RPN
        self.matcher = Matcher(rpn.FG_IOU_THRESHOLD, rpn.BG_IOU_THRESHOLD, True, rpn.YAW_THRESHOLD)
        self.sampler = BalancedPositiveNegativeSampler(rpn.BATCH_SIZE_PER_IMAGE, rpn.POSITIVE_FRACTION)
        ay, az = rpn.LABEL_AUG_THICKNESS_Y_TAR_ANC, rpn.LABEL_AUG_THICKNESS_Z_TAR_ANC
        self.aug = {'target_Y': ay[0], 'anchor_Y': ay[1], 'target_Z': az[0], 'anchor_Z': az[1]}
        self.yaw_loss_mode = cfg.MODEL.LOSS.YAW_MODE

    @torch.no_grad()
    def prepare_targets(self, anchors, gt_boxes):
        """-> labels fp32 [N] (1 pos, 0 neg, -1 ignored), regression targets [N,7]  (loss_3d.py:178-213)."""
        if gt_boxes.shape[0] == 0:
            return torch.zeros(anchors.shape[0], device=anchors.device), torch.zeros_like(anchors)
        q = box_ops.boxes_iou_3d(gt_boxes, anchors, self.aug, criterion=2, flag='rpn_label_generation')
        yaw_diff = torch.abs(box_ops.limit_period(gt_boxes[:, -1].view(-1, 1) - anchors[:, -1].view(1, -1), 0.5, math.pi))
        matched = self.matcher(q, yaw_diff=yaw_diff)
        labels = (matched >= 0).to(torch.float32)
        labels[matched == Matcher.BETWEEN_THRESHOLDS] = -1
        reg = box_encode(gt_boxes[matched.clamp(min=0)], anchors)
        return labels, reg

    def __call__(self, anchors, objectness, box_regression, gt_boxes):
        labels, reg_targets = self.prepare_targets(anchors, gt_boxes)
        pos, neg = self.sampler(labels)
        sampled = torch.cat([pos, neg], dim=0)
        box_loss = smooth_l1_loss(box_regression[pos], reg_targets[pos], anchors[pos], beta=1.0 / 9,
                                  size_average=False, yaw_loss_mode=self.yaw_loss_mode) / max(sampled.numel(), 1)
        obj_loss = F.binary_cross_entropy_with_logits(objectness[sampled], labels[sampled])
        return obj_loss, box_loss


class ROILoss(object):
    def __init__(self, cfg):
        rh = cfg.MODEL.ROI_HEADS
        self.matcher = Matcher(rh.FG_IOU_THRESHOLD, rh.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
        self.sampler = BalancedPositiveNegativeSampler(rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION)
        ay, az = rh.LABEL_AUG_THICKNESS_Y_TAR_ANC, rh.LABEL_AUG_THICKNESS_Z_TAR_ANC
        self.aug = {'target_Y': ay[0], 'anchor_Y': ay[1], 'target_Z': az[0], 'anchor_Z': az[1]}
        self.weights = tuple(rh.BBOX_REG_WEIGHTS)
        self.yaw_loss_mode = cfg.MODEL.LOSS.YAW_MODE

    @torch.no_grad()
    def subsample(self, proposals, gt_boxes, gt_labels):
        """-> (sampled proposals [S,7], labels int64 [S], regression targets [S,7])  (loss.py:66-160)."""
        if gt_boxes.shape[0] == 0:
            labels = torch.zeros(proposals.shape[0], dtype=torch.int64, device=proposals.device)
            reg = torch.zeros_like(proposals)
        else:
            q = box_ops.boxes_iou_3d(gt_boxes, proposals, self.aug, criterion=-1, flag='roi_label_generation')
            matched = self.matcher(q)
            labels = gt_labels[matched.clamp(min=0)].to(torch.int64)
            labels[matched == Matcher.BELOW_LOW_THRESHOLD] = 0
            labels[matched == Matcher.BETWEEN_THRESHOLDS] = -1
            reg = box_encode(gt_boxes[matched.clamp(min=0)], proposals, self.weights)
        pos, neg = self.sampler(labels)
        keep = torch.sort(torch.cat([pos, neg]))[0]             # nonzero(pos | neg) order
        return proposals[keep], labels[keep], reg[keep]

    def __call__(self, class_logits, box_regression, proposals, labels, reg_targets):
        cls_loss = F.cross_entropy(class_logits, labels)
        pos = torch.nonzero(labels > 0).squeeze(1)
        cols = 7 * labels[pos][:, None] + torch.arange(7, device=labels.device)
        box_loss = smooth_l1_loss(box_regression[pos[:, None], cols], reg_targets[pos], proposals[pos],
                                  size_average=False, beta=1 / 5., yaw_loss_mode=self.yaw_loss_mode)
        return cls_loss, box_loss / max(labels.numel(), 1)


# ----------------------------------------------------------------------------------------------
def make_optimizer(cfg, model):
    """solver/build.py:7-20: bias parameters get lr x BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS.  The reference makes one
    group per parameter; parameters with equal hyper-parameters are put in ONE group here (same update rule per
    parameter), so that the foreach SGD kernels cover ~100 tensors per launch instead of one."""
    groups = {}
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
        groups.setdefault((lr, wd), []).append(value)
    params = [{"params": v, "lr": lr, "weight_decay": wd} for (lr, wd), v in groups.items()]
    return torch.optim.SGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    """solver/lr_scheduler.py: linear warm-up from WARMUP_FACTOR, then x GAMMA at each milestone (iterations)."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        self.milestones, self.gamma = sorted(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        f = 1.0
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                f = self.warmup_factor
            else:
                alpha = self.last_epoch / self.warmup_iters
                f = self.warmup_factor * (1 - alpha) + alpha
        return [b * f * self.gamma ** bisect_right(self.milestones, self.last_epoch) for b in self.base_lrs]


def make_lr_scheduler(cfg, optimizer, examples_per_epoch):
    """solver/build.py:23-35: epochs -> iterations with INPUT.Example_num (= examples_per_epoch, set by the data
    loader in the reference); true division before int(), warm-up capped at 500 iterations (:27)."""
    per = examples_per_epoch / cfg.SOLVER.IMS_PER_BATCH
    steps = [int(e * per) for e in cfg.SOLVER.LR_STEP_EPOCHS]
    warmup = min(int(cfg.SOLVER.WARMUP_EPOCHS * per), 500)
    return WarmupMultiStepLR(optimizer, steps, cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR, warmup,
                             cfg.SOLVER.WARMUP_METHOD)


def freeze_unused(model):
    """requires_grad = False for the parameters no loss depends on (FPN_Net.unused_modules): with them frozen every
    remaining parameter gets a gradient in every step, so DistributedDataParallel needs no find_unused_parameters
    graph walk.  -> number of parameters frozen."""
    backbone = getattr(model, "backbone", None)
    n = 0
    if backbone is not None and hasattr(backbone, "unused_modules"):
        for mod in backbone.unused_modules():
            for p in mod.parameters():
                if p.requires_grad:
                    p.requires_grad_(False)
                    n += p.numel()
    return n


def wrap_ddp(model, local_rank=None, freeze=True):
    """tools/train_net_sparse3d.py:52-57: DistributedDataParallel without buffer broadcast (BatchNorm statistics are per
    forward).  The top-down levels that no RPN / RoI map consumes never receive gradients: they are
    frozen (`freeze`), otherwise DDP searches for them every step (find_unused_parameters).  Backend "nccl" = RCCL:
    the bucketed all-reduce of ~137 MB of fp32 gradients overlaps the backward pass."""
    frozen = freeze_unused(model) if freeze else 0
    kw = dict(broadcast_buffers=False, find_unused_parameters=not (freeze and frozen > 0) and _has_unused(model))
    if local_rank is not None:
        kw.update(device_ids=[local_rank], output_device=local_rank)
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


def _has_unused(model):
    backbone = getattr(model, "backbone", None)
    if backbone is None or not hasattr(backbone, "unused_modules"):
        return True      # unknown module: let DDP look
    return any(p.requires_grad for mod in backbone.unused_modules() for p in mod.parameters())


def reduce_loss_dict(loss_dict):
    """maskrcnn_benchmark/engine/trainer_sparse3d.py:17-39: the 4-12 loss scalars summed onto rank 0 (one small
    `reduce`), which divides by the world size; other ranks get their partial sums back, as in the reference."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        all_losses = torch.stack([loss_dict[k].detach() for k in names], dim=0)
        dist.reduce(all_losses, dst=0)
        if dist.get_rank() == 0:
            all_losses /= world
        return {k: v for k, v in zip(names, all_losses)}


def train_step(model, optimizer, scheduler, points, targets):
    """One iteration of do_train (engine/trainer_sparse3d.py:84-123): forward -> sum of the losses -> backward (DDP
    all-reduces the gradients bucket by bucket meanwhile) -> SGD step -> LR schedule.  -> (loss dict of this rank,
    reduced loss dict for logging)."""
    loss_dict = model(points, targets)
    losses = sum(loss_dict.values())
    reduced = reduce_loss_dict(loss_dict)
    optimizer.zero_grad()
    losses.backward()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss_dict, reduced

"""Generates tests/golden/ref_python.npz by importing the importable pure-torch pieces of the
reference from /root/reference (run in the build container only; the reference does not travel to
the GPU box).  Also copies the reference's demo detections (data files) used as realistic box
distributions:  demo/suncg_test_5_iou_3_augth_2/text_models/room_*.txt  ->  tests/golden/rooms.npz

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from utils3d.geometric_torch import limit_period, OBJ_DEF            # noqa: E402
from maskrcnn_benchmark.structures.bounding_box_3d import BoxList3D   # noqa: E402
from maskrcnn_benchmark.modeling.matcher import Matcher               # noqa: E402

torch.manual_seed(1234)
out = {}
# utils3d/geometric_torch.py:4-10
val = (torch.rand(4096) - 0.5) * 20
out["lp_in"] = val.numpy()
out["lp_half"] = limit_period(val, 0.5, math.pi).numpy()
out["lp_zero"] = limit_period(val, 0.0, math.pi).numpy()
# structures/bounding_box_3d.py:221-242 yx_zb -> standard (+ limit_yaw in the constructor)
b = torch.rand(512, 7)
b[:, 0:2] = b[:, 0:2] * 20
b[:, 2] = b[:, 2] * 0.5
b[:, 3] = 0.05 + b[:, 3] * 0.4
b[:, 4] = b[:, 3] + 0.2 + b[:, 4] * 5
b[:, 5] = 0.5 + b[:, 5] * 2.5
b[:, 6] = (b[:, 6] - 0.5) * 3.0
bl = BoxList3D(b.clone(), None, "yx_zb", None, {"prediction": True})
out["conv_yxzb"] = bl.bbox3d.numpy().copy()
out["conv_standard"] = bl.convert("standard").bbox3d.numpy().copy()
# modeling/matcher.py:58-177: RPN-style (low-quality matches + ignore-nearby + yaw mask) and ROI-style
iou = torch.rand(37, 900) ** 3
out["match_iou"] = iou.numpy()
out["match_res"] = Matcher(0.55, 0.2, allow_low_quality_matches=True, yaw_threshold=math.pi)(iou.clone()).numpy()
iou2 = (torch.rand(23, 4000) ** 8) * 0.9
iou2[torch.arange(23), torch.randint(0, 4000, (23,))] = 0.3 + 0.6 * torch.rand(23)
yaw = torch.rand(23, 4000) * 1.5
out["match2_iou"], out["match2_yaw"] = iou2.numpy(), yaw.numpy()
out["match2_rpn"] = Matcher(0.55, 0.2, allow_low_quality_matches=True, yaw_threshold=0.7)(iou2.clone(), yaw_diff=yaw).numpy()
out["match2_roi"] = Matcher(0.5, 0.5, allow_low_quality_matches=False)(iou2.clone()).numpy()
np.savez_compressed(os.path.join(HERE, "ref_python.npz"), **out)

rooms = {}
d = os.path.join(REF, "demo/suncg_test_5_iou_3_augth_2/text_models")
for i in range(5):
    rooms[f"room_{i}"] = np.loadtxt(os.path.join(d, f"room_{i}.txt")).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "rooms.npz"), **rooms)
print({k: v.shape for k, v in out.items()}, {k: v.shape for k, v in rooms.items()})

# ---- round 2: the other reference modules that import cleanly in the build container ------------------------------
from types import SimpleNamespace                                                         # noqa: E402
from maskrcnn_benchmark.modeling.seperate_classifier import SeperateClassifier            # noqa: E402
from maskrcnn_benchmark.solver.build import make_lr_scheduler                             # noqa: E402
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler  # noqa: E402
import importlib.util                                                                     # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_sl1", os.path.join(REF, "maskrcnn_benchmark/layers/smooth_l1_loss.py"))
ref_sl1 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_sl1)

out2 = {}
torch.manual_seed(4321)
# modeling/seperate_classifier.py:7-55,268-297 for 3G6c (SEPARATE_CLASSES [['wall'], ['ceiling','floor']] -> ids [[1],[5,4]],
# tools/train_net_sparse3d.py intact_cfg) and SeW4c ([['wall']] -> [[1]], 4 classes)
for tag, sep_ids, n_cls in (("3g6c", [[1], [5, 4]], 6), ("sew4c", [[1]], 4)):
    sc = SeperateClassifier([list(s) for s in sep_ids], n_cls)
    out2[f"sep_{tag}_ids"] = np.array([c for g in sep_ids for c in g] + [-1] + [len(g) for g in sep_ids], np.int64)
    out2[f"sep_{tag}_grouped_flat"] = np.array([c for g in sc.grouped_classes for c in g], np.int64)
    out2[f"sep_{tag}_class_nums"] = np.array(sc.class_nums, np.int64)
    out2[f"sep_{tag}_total"] = np.array([sc.seperated_num_classes_total, sc.group_num], np.int64)
    out2[f"sep_{tag}_org_to_sep"] = sc.org_labels_to_sep_labels.numpy().astype(np.int64)
    labels = torch.randint(1, n_cls, (57,))
    boxes = torch.rand(57, 7)
    bl = BoxList3D(boxes.clone(), None, "yx_zb", None, {"prediction": False})
    bl.add_field("labels", labels.clone())
    tg = sc.seperate_targets_and_update_labels([bl])
    out2[f"sep_{tag}_labels_in"], out2[f"sep_{tag}_boxes_in"] = labels.numpy(), bl.bbox3d.numpy().copy()
    for gi, t in enumerate(tg):
        out2[f"sep_{tag}_tg{gi}_boxes"] = t[0].bbox3d.numpy().copy()
        out2[f"sep_{tag}_tg{gi}_labels"] = t[0].get_field("labels").numpy().astype(np.int64)
    # seperate_pred_logits / seperate_pred_box (:222-239) on random head outputs with random group ids
    n = 41
    sep_id = torch.randint(0, sc.group_num, (n,))
    ids_g = [torch.nonzero(sep_id == gi).view(-1) for gi in range(sc.group_num)]
    logits = torch.randn(n, sc.seperated_num_classes_total)
    reg = torch.randn(n, sc.seperated_num_classes_total * 7)
    out2[f"sep_{tag}_sep_id"], out2[f"sep_{tag}_logits"], out2[f"sep_{tag}_reg"] = sep_id.numpy(), logits.numpy(), reg.numpy()
    for gi, (lg, rg) in enumerate(zip(sc.seperate_pred_logits(logits, ids_g), sc.seperate_pred_box(reg, ids_g))):
        out2[f"sep_{tag}_logits_g{gi}"], out2[f"sep_{tag}_reg_g{gi}"] = lg.numpy(), rg.numpy()
    for gi in range(sc.group_num):      # label map back to the original ids (post_processor :312)
        out2[f"sep_{tag}_back_g{gi}"] = sc.sep_labels_to_org_labels[gi].numpy().astype(np.int64)

# solver/build.py:23-35 + solver/lr_scheduler.py for the three yamls (BASE_LR, LR_STEP_EPOCHS, WARMUP_EPOCHS as in
# configs/4c/4c_Fpn432_bs1_lr5_SD.yaml, configs/6c/6c_Fpn4321_bs1_lr5.yaml, configs/3G6c/3G6c_Fpn4321_bs1_lr2.yaml;
# GAMMA / WARMUP_FACTOR / WARMUP_METHOD = config/defaults.py), two dataset sizes each
probe_its = {}
for tag, base_lr, steps, warm in (("4c", 0.005, (100,), 1.0), ("6c", 0.005, (5,), 0.1), ("3g6c", 0.002, (10,), 0.1)):
    for n_ex in (300, 7007):
        cfg_ns = SimpleNamespace(INPUT=SimpleNamespace(Example_num=n_ex),
                                 SOLVER=SimpleNamespace(IMS_PER_BATCH=1, LR_STEP_EPOCHS=steps, WARMUP_EPOCHS=warm,
                                                        GAMMA=0.1, WARMUP_FACTOR=1.0 / 3, WARMUP_METHOD="linear"))
        lin = torch.nn.Linear(2, 2)
        opt = torch.optim.SGD([{"params": [lin.weight], "lr": base_lr}, {"params": [lin.bias], "lr": base_lr * 2}],
                              base_lr, momentum=0.9)
        sched = make_lr_scheduler(cfg_ns, opt)
        total = int(steps[0] * n_ex) + 40
        its = sorted(set(list(range(0, 620)) + list(range(max(0, int(steps[0] * n_ex) - 20), total))))
        want = set(its)
        vals = []
        for it in range(total):
            if it in want:
                vals.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sched.step()
        out2[f"lr_{tag}_{n_ex}_its"], out2[f"lr_{tag}_{n_ex}_vals"] = np.array(its, np.int64), np.array(vals, np.float64)

# modeling/balanced_positive_negative_sampler.py under a fixed CPU seed: the selected masks (membership) for an RPN-like
# label vector (few positives), a RoI-like one (many positives) and the degenerate cases
for tag, (npos, nneg, nign, bs, frac) in {"rpn": (37, 30000, 900, 256, 0.5), "roi": (400, 500, 100, 512, 0.25),
                                         "fewneg": (300, 40, 0, 256, 0.5), "nopos": (0, 1000, 10, 256, 0.5)}.items():
    lab = torch.cat([torch.randint(1, 6, (npos,)).float(), torch.zeros(nneg), -torch.ones(nign)])
    lab = lab[torch.randperm(lab.numel())]
    torch.manual_seed(99)
    pos, neg = BalancedPositiveNegativeSampler(bs, frac)([lab])
    out2[f"samp_{tag}_labels"] = lab.numpy()
    out2[f"samp_{tag}_cfg"] = np.array([bs, frac], np.float64)
    out2[f"samp_{tag}_pos"], out2[f"samp_{tag}_neg"] = pos[0].numpy().astype(np.uint8), neg[0].numpy().astype(np.uint8)

# layers/smooth_l1_loss.py:32-49 (+ get_yaw_loss :15-30): Diff, SinDiff and weighted SinDiff, both reductions
torch.manual_seed(7)
inp, tgt = torch.randn(200, 7) * 0.5, torch.randn(200, 7) * 0.5
# the function asserts `anchor.shape` (:37) and reads `anchor.bbox3d` (:27): a tensor that also answers `.bbox3d`
anc = torch.cat([torch.rand(200, 6) + 0.1, (torch.rand(200, 1) - 0.5) * 3.5], 1)
anc.bbox3d = anc
out2["sl1_in"], out2["sl1_tgt"], out2["sl1_anchor"] = inp.numpy(), tgt.numpy(), anc.numpy().copy()
for mode in ("Diff", "SinDiff", "SinDiff_2.5"):
    for beta, avg in ((1.0 / 9, False), (1.0 / 5, False), (1.0 / 9, True)):
        out2[f"sl1_{mode}_{beta:.3f}_{int(avg)}"] = np.array(
            ref_sl1.smooth_l1_loss(inp.clone(), tgt.clone(), anc, beta=beta, size_average=avg, yaw_loss_mode=mode).item())
np.savez_compressed(os.path.join(HERE, "ref_python2.npz"), **out2)
print("ref_python2:", len(out2), "arrays")

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

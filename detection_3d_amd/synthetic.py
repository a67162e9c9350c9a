"""Synthetic SYNBIM-shaped scenes and box sets.

The dataset itself is not available (reference README.md:11), so bench.py and the parity
tests use this generator: points on wall / floor / ceiling planes of a 25 x 19 x 2.7 m
building, 9 feature channels (xyz, colour, normal) as in
data3d/suncg_utils/suncg_dataset.py:146-149.
"""
import numpy as np


def make_scene(seed=0, n_points=500_000, extent=(25.0, 19.0, 2.7)):
    """Returns pcl float32 [N, 9] (xyz, rgb, normal) in metres."""
    rng = np.random.RandomState(seed)
    ex, ey, ez = extent
    n = n_points
    kind = rng.choice(4, size=n, p=[0.3, 0.3, 0.2, 0.2])
    xyz = rng.rand(n, 3) * np.array([ex, ey, ez])
    nrm = np.zeros((n, 3), np.float64)
    xs = np.linspace(0.05, ex - 0.05, 6)
    ys = np.linspace(0.05, ey - 0.05, 4)
    m = kind == 0
    xyz[m, 0] = xs[rng.randint(0, len(xs), m.sum())]
    nrm[m, 0] = 1
    m = kind == 1
    xyz[m, 1] = ys[rng.randint(0, len(ys), m.sum())]
    nrm[m, 1] = 1
    m = kind == 2
    xyz[m, 2] = 0.02
    nrm[m, 2] = 1
    m = kind == 3
    xyz[m, 2] = ez - 0.02
    nrm[m, 2] = -1
    rgb = rng.rand(n, 3)
    return np.concatenate([xyz, rgb, nrm], 1).astype(np.float32)


def make_boxes(seed=0, n=2000, extent=(25.0, 19.0, 2.7)):
    """Wall-like boxes in yx_zb mode [xc, yc, z_bot, dy(thickness), dx(length), dz, yaw] + scores."""
    rng = np.random.RandomState(seed)
    ex, ey, ez = extent
    b = np.zeros((n, 7), np.float64)
    b[:, 0] = rng.rand(n) * ex
    b[:, 1] = rng.rand(n) * ey
    b[:, 2] = rng.rand(n) * 0.3
    b[:, 3] = 0.08 + rng.rand(n) * 0.22
    b[:, 4] = 0.5 + rng.rand(n) * 5.5
    b[:, 5] = 1.0 + rng.rand(n) * 1.8
    base = np.array([0, np.pi / 2, -np.pi / 2, np.pi / 4, -np.pi / 4])
    b[:, 6] = base[rng.randint(0, 5, n)] + rng.randn(n) * 0.02
    scores = rng.rand(n)
    return b.astype(np.float32), scores.astype(np.float32)


def make_targets(seed=0, extent=(25.0, 19.0, 2.7)):
    """Ground-truth boxes of the synthetic building of make_scene: one wall per x / y plane (yx_zb mode:
    xc, yc, z_bot, thickness, length, height, yaw; yaw 0 = length along x) plus random windows and doors on
    them.  Labels follow data3d/suncg_utils/suncg_metas.py: wall 1, window 2, door 3."""
    rng = np.random.RandomState(seed + 12345)
    ex, ey, ez = extent
    xs = np.linspace(0.05, ex - 0.05, 6)
    ys = np.linspace(0.05, ey - 0.05, 4)
    boxes, labels = [], []
    for x in xs:
        boxes.append([x, ey / 2, 0.0, 0.1, ey, ez, -np.pi / 2])
        labels.append(1)
    for y in ys:
        boxes.append([ex / 2, y, 0.0, 0.1, ex, ez, 0.0])
        labels.append(1)
    for _ in range(12):
        if rng.rand() < 0.5:
            x = xs[rng.randint(len(xs))]
            c = [x, rng.uniform(1, ey - 1), 0.0, 0.12, 0.0, 0.0, -np.pi / 2]
        else:
            y = ys[rng.randint(len(ys))]
            c = [rng.uniform(1, ex - 1), y, 0.0, 0.12, 0.0, 0.0, 0.0]
        if rng.rand() < 0.5:      # window
            c[2], c[4], c[5] = 0.9, rng.uniform(0.8, 1.8), rng.uniform(0.8, 1.4)
            labels.append(2)
        else:                     # door
            c[2], c[4], c[5] = 0.0, rng.uniform(0.8, 1.2), rng.uniform(1.9, 2.2)
            labels.append(3)
        boxes.append(c)
    return np.asarray(boxes, np.float32), np.asarray(labels, np.int64)


def yx_zb_to_standard(boxes):
    """Inverse of scene_io.standard_to_yx_zb (utils3d/bbox3d_ops.py:158-176): yx_zb (xc, yc, z_bot, dy, dx, dz, yaw)
    -> the dataset files' "standard" mode (xc, yc, zc, x_size, y_size, z_size, yaw in [0, pi))."""
    b = np.array(boxes, dtype=np.float32).reshape(-1, 7)[:, [0, 1, 2, 4, 3, 5, 6]]
    b[:, 2] = b[:, 2] + b[:, 5] * 0.5
    yaw = b[:, 6] + np.float32(np.pi * 0.5)
    b[:, 6] = yaw - np.floor(yaw / np.pi) * np.pi
    return b


def write_scene_file(path, seed=0, n_points=500_000, classes=('background', 'wall', 'window', 'door')):
    """One synthetic building in the dataset's file format (scene_io.save_scene: `.npz`, or the reference's `.pth`):
    the point cloud of make_scene and the boxes of make_targets per class name in standard mode."""
    from .scene_io import save_scene
    pcl = make_scene(seed, n_points)
    boxes, labels = make_targets(seed)
    per_class = {}
    for l in np.unique(labels):
        per_class[classes[int(l)]] = yx_zb_to_standard(boxes[labels == l])
    save_scene(path, pcl, per_class)
    return path

"""Shared test helpers: small synthetic scenes and a numpy composition of the FPN backbone out of
oracle ops (the checker for the HIP backbone)."""
import numpy as np

import oracle


def small_scene(seed, n_points, extent, full_scale, scale=50):
    from detection_3d_amd.synthetic import make_scene
    pcl = make_scene(seed, n_points, extent)
    coords, feats = oracle.voxelize(pcl, scale, full_scale)
    return pcl, coords, feats


def canon_rules(trip):
    """sorted (offset, in, out) rows of an (in, out, offset) triple list"""
    t = np.asarray(trip, np.int64).reshape(-1, 3)
    order = np.lexsort((t[:, 1], t[:, 0], t[:, 2]))
    return t[order]


def nbr_to_rules(nbr):
    n, K = nbr.shape
    out_ids, ks = np.nonzero(nbr >= 0)
    return np.stack([nbr[out_ids, ks], out_ids, ks], 1)


from oracle.detector_port import OracleDetector, OracleFPN  # noqa: E402,F401


def sort_by_loc(feats, loc):
    loc = np.asarray(loc)
    order = np.lexsort((loc[:, 2], loc[:, 1], loc[:, 0], loc[:, 3]))
    return feats[order], loc[order]


def label_differences_sit_on_thresholds(lab_a, lab_b, q_a, q_b, low, high, eps=1e-6):
    """Matcher labels from two IoU matrices q_a / q_b [M gt, N] (both already yaw-masked) may differ only where an IoU sits on
    one of the Matcher's thresholds (modeling/matcher.py:84-100,131-158): the low / high thresholds on a column's
    maximum, a gt's best quality (the ties that become low-quality matches) and its ignore threshold max(0.02,
    best - 0.05).  -> (number of differing columns, list of unexplained column indices)."""
    lab_a, lab_b = np.asarray(lab_a), np.asarray(lab_b)
    diff = np.nonzero(lab_a != lab_b)[0]
    bad = []
    if not len(diff):
        return 0, bad
    q_a, q_b = np.asarray(q_a, np.float64), np.asarray(q_b, np.float64)
    best = np.stack([q_a.max(1), q_b.max(1)])                                   # [2, M]
    ign = np.maximum(0.02, best - 0.05)
    for j in diff:
        ok = False
        for q in (q_a, q_b):
            col = q[:, j]
            m = col.max()
            ok |= min(abs(m - low), abs(m - high)) <= eps
            ok |= bool((np.abs(col[None] - best) <= eps).any()) or bool((np.abs(col[None] - ign) <= eps).any())
        if not ok:
            bad.append(int(j))
    return len(diff), bad

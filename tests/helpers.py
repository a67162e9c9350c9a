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

"""Scene files and prefetcher: safe loading of the reference's `.pth` layout, the
standard -> yx_zb conversion pinned by the reference's own BoxList3D.convert vectors, rank sharding."""
import os
import pickle

import numpy as np
import pytest
import torch

from detection_3d_amd import scene_io

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _scene(seed, n=2000):
    rng = np.random.RandomState(seed)
    pcl = rng.rand(n, 9).astype(np.float32)
    pcl[:, 0:3] = pcl[:, 0:3] * [8, 6, 2.7] + [3.0, -2.0, 0.5]
    walls = np.array([[5, 1, 1.85, 4.0, 0.1, 2.7, 0.0], [7, 0, 1.85, 6.0, 0.1, 2.7, np.pi / 2]], np.float32)
    wins = np.array([[5, 1, 1.5, 1.2, 0.12, 1.0, 0.0]], np.float32)
    return pcl, {"wall": walls, "window": wins, "floor": np.array([[7, 1, 0.45, 8, 6, 0.1, 0.0]], np.float32)}


def test_pth_roundtrip_with_safe_loader(tmp_path):
    pcl, boxes = _scene(0)
    f = str(tmp_path / "pcl_0.pth")
    scene_io.save_scene(f, pcl, boxes)                       # == torch.save((pcl, boxes)) of indoor_data_util.py:185
    p2, b2 = scene_io.load_scene(f)
    assert np.array_equal(p2, pcl) and set(b2) == set(boxes)
    for k in boxes:
        assert np.array_equal(b2[k], boxes[k])
    g = str(tmp_path / "pcl_0t.pth")                         # boxes saved as tensors (suncg_dataset.py:93-95)
    torch.save((pcl, {k: torch.from_numpy(v) for k, v in boxes.items()}), g)
    p3, b3 = scene_io.load_scene(g)
    assert np.array_equal(p3, pcl) and np.array_equal(b3["wall"], boxes["wall"])
    h = str(tmp_path / "pcl_0.npz")
    scene_io.save_scene(h, pcl, boxes)
    p4, b4 = scene_io.load_scene(h)
    assert np.array_equal(p4, pcl) and np.array_equal(b4["window"], boxes["window"])


class _Evil(object):
    def __reduce__(self):
        return (os.system, ("echo should-never-run > /dev/null",))


def test_files_with_code_are_refused(tmp_path):
    f = str(tmp_path / "evil.pth")
    with open(f, "wb") as fh:
        pickle.dump((np.zeros((1, 9), np.float32), {"wall": _Evil()}), fh)
    with pytest.raises(RuntimeError, match="refused"):
        scene_io.load_scene(f)


def test_standard_to_yx_zb_inverts_reference_convert():
    g = np.load(os.path.join(GOLD, "ref_python.npz"))
    got = scene_io.standard_to_yx_zb(g["conv_standard"])
    want = g["conv_yxzb"]
    assert np.abs(got[:, :6] - want[:, :6]).max() < 2e-6
    d = np.abs(got[:, 6] - want[:, 6])
    assert np.minimum(d, np.abs(d - np.pi)).max() < 1e-5     # yaw is defined modulo pi
    assert (got[:, 6] >= -np.pi / 2 - 1e-6).all() and (got[:, 6] < np.pi / 2 + 1e-6).all()


def test_targets_follow_the_point_shift_and_class_order():
    pcl, boxes = _scene(1)
    classes = ['background', 'wall', 'door', 'window']
    tg = scene_io.scene_targets(pcl, boxes, classes, 50)
    assert tg["labels"].tolist() == [1, 1, 2]                # wall 1, window 2 (suncg_metas.py order), floor skipped
    off = -(pcl[:, 0:3].astype(np.float64) * 50).min(0) / 50
    want_c = boxes["wall"][0, 0:3] + off
    want_c[2] -= boxes["wall"][0, 5] * 0.5                   # centre z -> bottom z
    assert np.allclose(tg["bbox3d"][0, 0:3], want_c, atol=1e-5)
    assert np.allclose(tg["bbox3d"][0, 3:6], [0.1, 4.0, 2.7])
    assert abs(tg["bbox3d"][0, 6] + np.pi / 2) < 1e-6 and abs(tg["bbox3d"][1, 6]) < 1e-6
    z = scene_io.set_yaw_zero(np.array([[0, 0, 0, 1, 3, 1, np.pi / 2]], np.float32))
    assert z[0, 3:5].tolist() == [3, 1] and z[0, 6] == 0


def test_prefetcher_shards_files_by_rank(tmp_path):
    files = []
    for i in range(5):
        pcl, boxes = _scene(10 + i, 300 + i)
        f = str(tmp_path / f"pcl_{i}.npz")
        scene_io.save_scene(f, pcl, boxes)
        files.append(f)
    classes = ['background', 'wall', 'door', 'window']
    seen = []
    for rank in range(2):
        pf = scene_io.ScenePrefetcher(files, classes, 50, device=None, rank=rank, world=2, depth=2, element_ids=range(9))
        got = [(p.shape[0], os.path.basename(path), t["labels"].shape[0]) for p, t, path in pf]
        assert [g[1] for g in got] == [f"pcl_{i}.npz" for i in range(rank, 5, 2)]
        assert all(g[0] == 300 + int(g[1][4]) and g[2] == 3 for g in got)
        seen += [g[1] for g in got]
    assert sorted(seen) == sorted(os.path.basename(f) for f in files)
    with pytest.raises(RuntimeError):
        list(scene_io.ScenePrefetcher([str(tmp_path / "pcl_0.npz"), __file__ + ".pth"], classes, 50))


@pytest.mark.gpu
def test_prefetcher_to_gpu_matches_direct_upload(tmp_path, dev):
    from detection_3d_amd.voxelize import voxelize
    files = []
    for i in range(3):
        pcl, boxes = _scene(20 + i, 5000)
        f = str(tmp_path / f"pcl_{i}.pth")
        scene_io.save_scene(f, pcl, boxes)
        files.append((f, pcl))
    classes = ['background', 'wall', 'door', 'window']
    pf = scene_io.ScenePrefetcher([f for f, _ in files], classes, 50, device=dev, depth=2)
    for (p, t, path), (f, pcl) in zip(pf, files):
        assert path == f and p.is_cuda and t["bbox3d"].is_cuda
        c1, f1 = voxelize(p, 50, [4096, 4096, 512])
        c2, f2 = voxelize(torch.from_numpy(pcl).to(dev), 50, [4096, 4096, 512])
        assert torch.equal(c1, c2) and torch.equal(f1, f2)


def test_load_scene_into_reads_the_stored_array_in_place(tmp_path):
    """The prefetcher's read path: the float32 point cloud of an uncompressed .npz lands in the caller's buffer by one
    readinto and equals load_scene's array; compressed / .pth scenes take the copying fallback with the same result."""
    pcl, boxes = _scene(31, 4000)
    f = str(tmp_path / "a.npz")
    scene_io.save_scene(f, pcl, boxes)
    assert scene_io._npz_stored_array(f, "pcl") is not None
    bufs = []

    def take(n):
        bufs.append(np.full(n + 7, -1.0, np.float32))
        return bufs[-1]

    got, b2 = scene_io.load_scene_into(f, take)
    want, b1 = scene_io.load_scene(f)
    assert np.array_equal(got, want) and got.base is not None and np.shares_memory(got, bufs[-1])
    assert sorted(b1) == sorted(b2) and all(np.array_equal(b1[k], b2[k]) for k in b1)
    g = str(tmp_path / "c.npz")
    np.savez_compressed(g, pcl=pcl, **{"box_" + k: v for k, v in boxes.items()})
    assert scene_io._npz_stored_array(g, "pcl") is None
    got2, _ = scene_io.load_scene_into(g, take)
    assert np.array_equal(got2, want)
    h = str(tmp_path / "d.pth")
    scene_io.save_scene(h, pcl, boxes)
    got3, _ = scene_io.load_scene_into(h, take)
    assert np.array_equal(got3, want)

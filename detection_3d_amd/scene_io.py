"""Scene I/O for the hot path: the per-building files of the reference and a prefetcher that
keeps the next buildings' point clouds in flight to the GPU while the current one is processed.

File format (data3d/indoor_data_util.py:171-185): `torch.save((pcl float32[N,9], {class: float32[M,7]}), 'pcl_i.pth')`,
boxes in "standard" mode (xc, yc, zc, x_size, y_size, z_size, yaw in [0, pi)).  The files are loaded with
`torch.load(weights_only=True)` plus an allow-list for numpy arrays, so nothing in a file is executed; a file the safe
loader refuses is reported, not unpickled.  `.npz` files with `pcl` and `box_<class>` arrays are accepted too (what
`save_scene` writes).

Host-side arithmetic follows data3d/suncg_utils/suncg_dataset.py:72-192 (`__getitem__` with its augmentations
disabled, as they are in the reference): boxes standard -> yx_zb (utils3d/bbox3d_ops.py:158-176), the per-axis minimum
of `xyz * scale` in float64 shifts points (on the GPU, d3d_voxelize) and boxes (here) alike.
"""
import os
import queue
import threading

import numpy as np
import torch

from .config import class_to_label

_ZERO_YAW_CLASSES = ('ceiling', 'floor', 'room')


def _numpy_safe_globals():
    """Globals a pickled numpy array needs; data only, no code from the file runs."""
    out = [np.ndarray, np.dtype]
    try:
        from numpy.core.multiarray import _reconstruct
        out.append(_reconstruct)
    except Exception:                                   # numpy >= 2 moved it
        from numpy._core.multiarray import _reconstruct
        out.append(_reconstruct)
    for name in ("Float32DType", "Float64DType", "Int64DType", "Int32DType", "UInt8DType", "BoolDType"):
        t = getattr(np.dtypes, name, None) if hasattr(np, "dtypes") else None
        if t is not None:
            out.append(t)
    return out


def load_scene(path):
    """-> (pcl float32 [N, F], {class: float32 [M, 7] standard boxes}).  Raises RuntimeError when the safe loader
    refuses the file."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as d:
            pcl = np.ascontiguousarray(d["pcl"], dtype=np.float32)
            boxes = {k[4:]: np.asarray(d[k], dtype=np.float32).reshape(-1, 7) for k in d.files if k.startswith("box_")}
        return pcl, boxes
    try:
        with torch.serialization.safe_globals(_numpy_safe_globals()):
            obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:                               # noqa: BLE001 - report, never fall back to unpickling
        raise RuntimeError(f"{path}: refused by torch.load(weights_only=True) ({type(e).__name__}: {e}); "
                           "re-export the scene with scene_io.save_scene") from e
    pcl, boxes = obj
    pcl = pcl.numpy() if isinstance(pcl, torch.Tensor) else np.asarray(pcl)
    out = {}
    for k, v in boxes.items():
        v = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)    # suncg_dataset.py:93-95
        out[k] = np.asarray(v, dtype=np.float32).reshape(-1, 7)
    return np.ascontiguousarray(pcl, dtype=np.float32), out


def save_scene(path, pcl, boxes):
    """Writes the `.npz` form (no pickle) or, for a `.pth` path, the reference's own `torch.save((pcl, boxes))`."""
    pcl = np.ascontiguousarray(pcl, dtype=np.float32)
    boxes = {k: np.asarray(v, dtype=np.float32).reshape(-1, 7) for k, v in boxes.items()}
    if path.endswith(".npz"):
        np.savez(path, pcl=pcl, **{"box_" + k: v for k, v in boxes.items()})
    else:
        torch.save((pcl, boxes), path)


def limit_period(val, offset, period):
    """utils3d/geometric_util.py limit_period: val - floor(val / period + offset) * period."""
    return val - np.floor(val / period + offset) * period


def standard_to_yx_zb(boxes):
    """Bbox3D.convert_to_yx_zb_boxes (utils3d/bbox3d_ops.py:158-176): swap the two horizontal sizes, centre z ->
    bottom z, yaw - pi/2 limited to [-pi/2, pi/2)."""
    b = np.array(boxes, dtype=np.float32).reshape(-1, 7)[:, [0, 1, 2, 4, 3, 5, 6]]
    b[:, 2] = b[:, 2] - b[:, 5] * 0.5
    b[:, 6] -= np.float32(np.pi * 0.5)
    b[:, 6] = limit_period(b[:, 6], 0.5, np.pi)
    return b


def set_yaw_zero(boxes):
    """Bbox3D.set_yaw_zero (utils3d/bbox3d_ops.py:178-195) for ceiling / floor / room."""
    b = np.array(boxes, dtype=np.float32).reshape(-1, 7)
    if b.shape[0] == 0:
        return b
    yaws = b[:, 6]
    assert np.mod(yaws, np.pi / 2).max() < 0.01 or np.mod(-yaws, np.pi / 2).max() < 0.01
    sw = np.abs(yaws / (np.pi / 2)).astype(np.int64)
    sy = b[:, 3] * (1 - sw) + b[:, 4] * sw
    sx = b[:, 4] * (1 - sw) + b[:, 3] * sw
    b[:, 3], b[:, 4], b[:, 6] = sy, sx, 0
    return b


def scene_targets(pcl, boxes_std, classes, scale):
    """Ground truth of one building in the detector's frame (suncg_dataset.py:97-166,235-250):
    -> {"bbox3d": float32 [M,7] yx_zb shifted like the points, "labels": int64 [M]}."""
    c2l = class_to_label(classes)
    a_min = (pcl[:, 0:3].astype(np.float64) * float(scale)).min(0) if pcl.shape[0] else np.zeros(3)
    offset = -a_min / float(scale)
    bb, ll = [], []
    for obj, b in boxes_std.items():
        if obj not in classes and 'all' not in classes:
            continue
        b = standard_to_yx_zb(b)
        if obj in _ZERO_YAW_CLASSES:
            b = set_yaw_zero(b)
        b[:, 0:3] += offset[None, :].astype(np.float64)            # float32 += float64, as the reference does
        bb.append(b)
        ll.append(np.full(b.shape[0], c2l[obj], dtype=np.int64))
        assert c2l[obj] > 0, "label 0 is background"
    if not bb:
        return {"bbox3d": np.zeros((0, 7), np.float32), "labels": np.zeros((0,), np.int64)}
    return {"bbox3d": np.concatenate(bb, 0).astype(np.float32), "labels": np.concatenate(ll, 0)}


def list_scene_files(root, scene_names):
    """suncg_dataset.py:52-56: every block file of the listed houses."""
    files = []
    for s in scene_names:
        d = os.path.join(root, "houses", s)
        files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".pth", ".npz")))
    return files


class ScenePrefetcher(object):
    """Iterates `(pcl, targets, path)` over `files[rank::world]` (one building per rank and step) with a
    reader thread that keeps `depth` buildings ahead: file read + box conversion on the host, pinned staging buffer,
    asynchronous copy on a side stream.  `pcl` is a float32 [N, F] tensor on `device` (host tensor when device is
    None), `targets` the dict of `scene_targets` as tensors on the same device.  The consumer's stream waits for the
    copy through an event, never the host."""

    def __init__(self, files, classes, scale, device=None, rank=0, world=1, depth=2, element_ids=None):
        self.files = list(files)[rank::world]
        self.classes, self.scale, self.device = list(classes), scale, device
        self.element_ids = None if element_ids is None else sorted(int(i) for i in element_ids)
        self.depth = max(1, int(depth))
        self._stream = torch.cuda.Stream(device=device) if device is not None else None

    def __len__(self):
        return len(self.files)

    def _load(self, path):
        pcl, boxes = load_scene(path)
        tg = scene_targets(pcl, boxes, self.classes, self.scale)
        if self.element_ids is not None:
            pcl = np.ascontiguousarray(pcl[:, self.element_ids])
        host = torch.from_numpy(pcl)
        tb, tl = torch.from_numpy(tg["bbox3d"]), torch.from_numpy(tg["labels"])
        if self.device is None:
            return host, {"bbox3d": tb, "labels": tl}, path, None
        host, tb, tl = host.pin_memory(), tb.pin_memory(), tl.pin_memory()
        with torch.cuda.stream(self._stream):
            dev = host.to(self.device, non_blocking=True)
            db, dl = tb.to(self.device, non_blocking=True), tl.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return dev, {"bbox3d": db, "labels": dl}, path, (ev, host, tb, tl)     # pinned sources stay alive until used

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def worker():
            try:
                for f in self.files:
                    if stop.is_set():
                        return
                    q.put(("ok", self._load(f)))
                q.put(("end", None))
            except BaseException as e:               # noqa: BLE001 - surfaced in the consumer
                q.put(("err", e))

        t = threading.Thread(target=worker, daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise item
                pcl, tg, path, keep = item
                if keep is not None:
                    torch.cuda.current_stream(self.device).wait_event(keep[0])
                    pcl.record_stream(torch.cuda.current_stream(self.device))
                yield pcl, tg, path
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)

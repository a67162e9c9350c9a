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


_ALLOWED = False


def _allow_numpy_globals():
    """Registers the numpy-array globals with torch's weights-only unpickler once, process-wide: the `safe_globals`
    context manager edits a shared list on entry and exit, so concurrent loads from reader threads would take each
    other's allow-list away."""
    global _ALLOWED
    if not _ALLOWED:
        torch.serialization.add_safe_globals(_numpy_safe_globals())
        _ALLOWED = True


def load_scene(path):
    """-> (pcl float32 [N, F], {class: float32 [M, 7] standard boxes}).  Raises RuntimeError when the safe loader
    refuses the file."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as d:
            pcl = np.ascontiguousarray(d["pcl"], dtype=np.float32)
            boxes = {k[4:]: np.asarray(d[k], dtype=np.float32).reshape(-1, 7) for k in d.files if k.startswith("box_")}
        return pcl, boxes
    try:
        _allow_numpy_globals()
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


def _npz_stored_array(path, name):
    """-> (file offset of the raw data, shape, dtype) of member `name`.npy of an UNCOMPRESSED .npz, or None.
    np.savez stores members uncompressed; their bytes can then be read straight into a caller's buffer with one
    readinto (which releases the GIL) instead of zipfile's chunked, CRC-checked, GIL-holding copy."""
    import struct
    import zipfile
    with zipfile.ZipFile(path) as z:
        try:
            info = z.getinfo(name + ".npy")
        except KeyError:
            return None
        if info.compress_type != zipfile.ZIP_STORED:
            return None
    with open(path, "rb") as f:
        f.seek(info.header_offset)
        hdr = f.read(30)
        if len(hdr) < 30 or hdr[:4] != b"PK\x03\x04":
            return None
        nlen, elen = struct.unpack("<HH", hdr[26:30])
        f.seek(info.header_offset + 30 + nlen + elen)
        version = np.lib.format.read_magic(f)
        read_header = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
        shape, fortran, dtype = read_header(f)
        if fortran or dtype.hasobject:
            return None
        return f.tell(), shape, dtype


def load_scene_into(path, take_buffer):
    """load_scene with the point cloud read straight into memory the caller provides: `take_buffer(n_floats)` returns a
    writable float32 numpy array of at least that many elements (e.g. a view of a pinned staging buffer).  Only for
    `.npz` scenes with an uncompressed float32 `pcl` (what save_scene writes); anything else falls back to load_scene
    plus one copy.  -> (pcl view [N, F] inside the buffer, boxes dict)."""
    span = _npz_stored_array(path, "pcl") if path.endswith(".npz") else None
    if span is None or span[2] != np.dtype(np.float32) or len(span[1]) != 2:
        pcl, boxes = load_scene(path)
        view = take_buffer(pcl.size)[:pcl.size].reshape(pcl.shape)
        view[...] = pcl
        return view, boxes
    offset, shape, _ = span
    n = int(shape[0]) * int(shape[1])
    view = take_buffer(n)[:n].reshape(shape)
    with open(path, "rb", buffering=0) as f:
        f.seek(offset)
        got = f.readinto(memoryview(view.reshape(-1)).cast("B"))
    if got != n * 4:
        raise RuntimeError(f"{path}: truncated point cloud ({got} of {n * 4} bytes)")
    with np.load(path, allow_pickle=False) as d:          # the boxes: a few hundred bytes per class
        boxes = {k[4:]: np.asarray(d[k], dtype=np.float32).reshape(-1, 7) for k in d.files if k.startswith("box_")}
    return view, boxes


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
    # min of (xyz * scale) in float64 = (min of xyz) * scale for scale > 0: the product is monotonic and the minimum's
    # product is computed from the same float32 value either way -- without a float64 copy of the whole cloud
    # (column by column: numpy's axis-0 reduction of a strided [N, 3] view is 4x slower than three 1-D ones)
    a_min = (np.array([pcl[:, d].min() for d in range(3)]).astype(np.float64) * float(scale)
             if pcl.shape[0] else np.zeros(3))
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
    """Iterates `(pcl, targets, path)` over `files[rank::world]` (one building per rank and step), in file
    order, with `workers` reader threads that keep up to `depth` buildings ahead: file read + box conversion on the host,
    copy into a REUSED pinned staging buffer (allocating pinned memory per building costs more than reading it),
    asynchronous copy to the device on a side stream.  `pcl` is a float32 [N, F] tensor on `device` (host tensor when
    device is None), `targets` the dict of `scene_targets` as tensors on the same device.  The consumer's stream waits for
    the copy through an event, never the host.  A 500 k-point building takes ~12 ms to read; three readers deliver one
    every ~5 ms, which is what a rank needs to keep its GPU busy at inference."""

    def __init__(self, files, classes, scale, device=None, rank=0, world=1, depth=6, element_ids=None, workers=3):
        self.files = list(files)[rank::world]
        self.classes, self.scale, self.device = list(classes), scale, device
        self.element_ids = None if element_ids is None else sorted(int(i) for i in element_ids)
        self.workers = max(1, min(int(workers), max(1, int(depth))))
        # a multiple of the workers: the buildings that share a staging slot (i, i + depth, ...) then belong to ONE reader,
        # which takes them in order -- no two readers ever wait for the same slot
        self.depth = -(-max(1, int(depth)) // self.workers) * self.workers
        self._stream = torch.cuda.Stream(device=device) if device is not None else None

    def __len__(self):
        return len(self.files)

    def _load_stage(self, path, pinned):
        """file -> (pinned) host memory -> device tensors.  With a device, the point cloud is read straight into the
        slot's pinned buffer (grown when a building is larger) and copied from there on the side stream."""
        if self.device is None:
            pcl, boxes = load_scene(path)
            tg = scene_targets(pcl, boxes, self.classes, self.scale)
            if self.element_ids is not None:
                pcl = np.ascontiguousarray(pcl[:, self.element_ids])
            return torch.from_numpy(pcl), {"bbox3d": torch.from_numpy(tg["bbox3d"]), "labels": torch.from_numpy(tg["labels"])}, None

        def take(n):
            if pinned.get("buf") is None or pinned["buf"].numel() < n:
                pinned["buf"] = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32).pin_memory()
            return pinned["buf"].numpy()

        pcl, boxes = load_scene_into(path, take)
        tg = scene_targets(pcl, boxes, self.classes, self.scale)
        host = torch.from_numpy(pcl)                       # a view of the pinned buffer
        if self.element_ids is not None:
            host = host[:, self.element_ids].contiguous().pin_memory()
            pinned["extra"] = host
        tb, tl = torch.from_numpy(tg["bbox3d"]), torch.from_numpy(tg["labels"])
        with torch.cuda.stream(self._stream):
            dev = host.to(self.device, non_blocking=True)
            db, dl = tb.to(self.device), tl.to(self.device)          # a few hundred bytes
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return dev, {"bbox3d": db, "labels": dl}, ev

    def __iter__(self):
        n, nw = len(self.files), self.workers
        slots = [dict() for _ in range(self.depth)]          # slot i % depth: pinned buffer + hand-over state
        ready = [threading.Event() for _ in range(n)]
        free = [threading.Semaphore(0) for _ in range(self.depth)]
        for sem in free:
            sem.release()
        results, errors = [None] * n, []
        stop = threading.Event()

        def worker(w):
            try:
                for i in range(w, n, nw):
                    slot = i % self.depth
                    while not free[slot].acquire(timeout=0.1):   # the building that used this slot has been consumed
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    results[i] = self._load_stage(self.files[i], slots[slot])
                    ready[i].set()
            except BaseException as e:                            # noqa: BLE001 - surfaced in the consumer
                errors.append(e)
                for ev in ready:
                    ev.set()

        threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(nw)]
        for t in threads:
            t.start()
        try:
            for i in range(n):
                ready[i].wait()
                if errors:
                    raise errors[0]
                pcl, tg, ev = results[i]
                results[i] = None
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    pcl.record_stream(cur)
                    ev.synchronize()                              # the pinned buffer may be overwritten from here on
                free[i % self.depth].release()
                yield pcl, tg, self.files[i]
        finally:
            stop.set()
            for t in threads:
                t.join(timeout=5)

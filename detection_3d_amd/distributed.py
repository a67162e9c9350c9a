"""Multi-GPU inference: one process per GPU, buildings sharded round-robin by rank, no collective on
the data path; detections are gathered to rank 0 as padded tensors.  Replaces the reference's
un-sharded loader (data3d/data.py:37-38) and its file-system gather (maskrcnn_benchmark/utils/comm.py:89-157).
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import torch
import torch.distributed as dist

MAX_DET = 200   # MODEL.ROI_HEADS.DETECTIONS_PER_IMG


def shard_scenes(n_scenes, rank, world_size):
    """Scene ids owned by `rank`: rank, rank + world, ..."""
    return list(range(rank, n_scenes, world_size))


def pack_detections(scene_id, result, max_det=MAX_DET):
    """-> fp32 [max_det + 1, 9] rows (x, y, z, dy, dx, dz, yaw, score, label); row 0 = (scene id, count, 0...).
    Raises when the building has more rows than max_det: nothing is dropped silently (callers size max_det from the
    data, `agree_capacity`)."""
    b, s, l = result["bbox3d"], result["scores"], result["labels"]
    n = int(b.shape[0])
    if n > max_det:
        raise ValueError(f"pack_detections: building {scene_id} has {n} rows, capacity {max_det}")
    out = torch.zeros((max_det + 1, 9), dtype=torch.float32, device=b.device)
    out[0, 0], out[0, 1] = float(scene_id), float(n)
    out[1:n + 1, 0:7] = b[:n]
    out[1:n + 1, 7] = s[:n]
    out[1:n + 1, 8] = l[:n].to(torch.float32)
    return out


def unpack_detections(packed):
    n = int(packed[0, 1].item())
    return int(packed[0, 0].item()), {"bbox3d": packed[1:n + 1, 0:7], "scores": packed[1:n + 1, 7],
                                      "labels": packed[1:n + 1, 8].to(torch.int64)}


def _collective_device(device):
    """where a small collective's tensor lives: the rank's GPU under RCCL ("nccl"), the host under gloo (CPU tests and
    the several-ranks-on-one-GPU rehearsals; gloo stages device tensors through the host anyway)"""
    if device is None or dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(device)


def agree_capacity(local_rows, floor, device=None):
    """Rows per building every rank packs with: max(floor, the largest row count any rank holds) -- one MAX all-reduce of
    one integer (RCCL / gloo), so that neither ground truth nor detections (d3d_post_select keeps every candidate tied at
    the cut, which can exceed DETECTIONS_PER_IMG) are ever truncated.  local_rows: iterable of this rank's row counts."""
    cap = max([int(floor)] + [int(v) for v in local_rows])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([cap], dtype=torch.int64, device=_collective_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap = int(t.item())
    return cap


def gather_detections(local, n_scenes, max_det=MAX_DET, device=None):
    """local: list of packed tensors of this rank's scenes.  Returns {scene_id: result} on rank 0 (None
    elsewhere).  Every rank contributes ceil(n_scenes / world) slots (empty slots have scene id -1).  device: where a
    rank WITHOUT buildings allocates its (empty) contribution -- under RCCL it must be the rank's GPU."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    slots = (n_scenes + world - 1) // world
    dev = local[0].device if local else (torch.device(device) if device is not None else torch.device("cpu"))
    buf = torch.zeros((slots, max_det + 1, 9), dtype=torch.float32, device=dev)
    buf[:, 0, 0] = -1
    for i, p in enumerate(local):
        buf[i] = p
    if world == 1:
        gathered = [buf]
    else:
        if dist.get_backend() == "gloo" and buf.is_cuda:
            buf = buf.cpu()
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
    if rank != 0:
        return None
    out = {}
    for g in gathered:
        for p in g:
            if p[0, 0].item() >= 0:
                sid, res = unpack_detections(p)
                out[sid] = res
    return out

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
    """-> fp32 [max_det + 1, 9] rows (x, y, z, dy, dx, dz, yaw, score, label); row 0 = (scene id, count, 0...)."""
    b, s, l = result["bbox3d"], result["scores"], result["labels"]
    n = min(int(b.shape[0]), max_det)
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


def gather_detections(local, n_scenes, max_det=MAX_DET):
    """local: list of packed tensors of this rank's scenes.  Returns {scene_id: result} on rank 0 (None
    elsewhere).  Every rank contributes ceil(n_scenes / world) slots (empty slots have scene id -1)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    slots = (n_scenes + world - 1) // world
    dev = local[0].device if local else torch.device("cpu")
    buf = torch.zeros((slots, max_det + 1, 9), dtype=torch.float32, device=dev)
    buf[:, 0, 0] = -1
    for i, p in enumerate(local):
        buf[i] = p
    if world == 1:
        gathered = [buf]
    else:
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

"""N>1 path on CPU: world-size-2 gloo processes shard the scene list, run a stand-in per-scene function
and gather padded detections on rank 0 (the same code bench.py / multi-GPU inference uses with RCCL)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_result(scene_id):
    g = torch.Generator().manual_seed(scene_id)
    n = 3 + scene_id % 5
    return {"bbox3d": torch.rand((n, 7), generator=g), "scores": torch.rand(n, generator=g),
            "labels": torch.randint(1, 4, (n,), generator=g)}


def _worker(rank, world, port, n_scenes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from detection_3d_amd.distributed import gather_detections, pack_detections, shard_scenes
    mine = shard_scenes(n_scenes, rank, world)
    local = [pack_detections(s, _fake_result(s)) for s in mine]
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's max-over-ranks timing
    res = gather_detections(local, n_scenes)
    if rank == 0:
        ok = sorted(res.keys()) == list(range(n_scenes)) and float(t.item()) == float(world)
        for s, r in res.items():
            w = _fake_result(s)
            ok = ok and torch.equal(r["bbox3d"], w["bbox3d"]) and torch.equal(r["labels"], w["labels"])
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    from detection_3d_amd.distributed import shard_scenes
    assert shard_scenes(7, 0, 2) == [0, 2, 4, 6] and shard_scenes(7, 1, 2) == [1, 3, 5]
    assert sorted(shard_scenes(5, 0, 8) + shard_scenes(5, 4, 8)) == [0, 4]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    [p.start() for p in procs]
    ok = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert ok and all(p.exitcode == 0 for p in procs)


def test_single_process_gather():
    from detection_3d_amd.distributed import gather_detections, pack_detections
    res = gather_detections([pack_detections(s, _fake_result(s)) for s in range(3)], 3)
    assert sorted(res) == [0, 1, 2] and res[2]["bbox3d"].shape[0] == 3 + 2

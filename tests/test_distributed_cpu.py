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


# ---- engine.train / engine.inference plumbing on world-size-2 gloo (the detector itself needs the GPU: a stand-in with
# ---- the same interface -- forward(points, targets) -> loss dict / detections, backbone.unused_modules() -- takes its place)
class _Backbone(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.used = torch.nn.Linear(9, 4)
        self.never = torch.nn.Linear(9, 4)          # like m_ups[5..7]: no loss depends on it

    def unused_modules(self):
        return [self.never]


class _StandIn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = _Backbone()
        self.head = torch.nn.Linear(4, 2)

    def forward(self, points, targets=None):
        coords, feats = points[0], points[1]
        h = self.head(torch.relu(self.backbone.used(feats)))
        if self.training:
            t = targets["bbox3d"][:, :2].mean(0) if targets["bbox3d"].shape[0] else torch.zeros(2)
            return {"loss_a": ((h.mean(0) - t) ** 2).sum(), "loss_b": h.abs().mean()}
        n = min(5, feats.shape[0])
        return {"bbox3d": feats[:n, :7].contiguous(), "scores": torch.linspace(0.9, 0.5, n), "labels": torch.ones(n, dtype=torch.int64)}


def _cpu_voxelize(pcl, cfg):
    return (pcl[:, :3] * 50).long(), pcl


def _engine_worker(rank, world, port, files, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from detection_3d_amd import engine, training as T
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.scene_io import ScenePrefetcher
    cfg = get_cfg("4c_Fpn432")
    ok = True
    # each rank reads files[rank::world]
    pre = ScenePrefetcher(files, cfg.INPUT.CLASSES, 50, device=None, rank=rank, world=world)
    ok = ok and [p for _, _, p in pre] == files[rank::world]
    # the loss reduce of trainer_sparse3d.py:17-39: rank 0 holds the mean, the others their partial sums
    red = T.reduce_loss_dict({"b": torch.tensor(2.0 + rank), "a": torch.tensor(1.0 * rank)})
    if rank == 0:
        ok = ok and abs(float(red["a"]) - 0.5) < 1e-6 and abs(float(red["b"]) - 2.5) < 1e-6
    torch.manual_seed(0)
    model = _StandIn()
    out = engine.train(model, cfg, files, None, steps=4, voxelize_fn=_cpu_voxelize)
    ok = ok and out["world"] == world and out["steps_timed"] == 3 and set(out["losses"]) == {"loss_a", "loss_b"}
    ok = ok and not any(p.requires_grad for p in model.backbone.never.parameters())        # frozen, not searched for
    ok = ok and all(p.grad is not None for p in model.parameters() if p.requires_grad)
    # after the same number of averaged-gradient steps every rank holds the same weights
    w = model.head.weight.detach().clone()
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok = ok and all(torch.allclose(ws[0], x) for x in ws)
    res = engine.inference(model, cfg, files, None, voxelize_fn=_cpu_voxelize)
    if rank == 0:
        dets, gts = res
        ok = ok and sorted(dets) == list(range(len(files))) and sorted(gts) == list(range(len(files)))
        ok = ok and all(d["bbox3d"].shape == (5, 7) for d in dets.values())
        ok = ok and all(g["bbox3d"].shape[0] == 22 and g["labels"].min() >= 1 for g in gts.values())
        q.put(bool(ok))
    else:
        ok = ok and res is None
        if not ok:
            q.put(False)
    dist.barrier()
    dist.destroy_process_group()


def test_engine_train_and_inference_world2_gloo(tmp_path):
    from detection_3d_amd.synthetic import write_scene_file
    files = [write_scene_file(str(tmp_path / f"s{i}.npz"), i, 2000) for i in range(5)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, files, q)) for r in range(2)]
    [p.start() for p in procs]
    ok = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert ok and all(p.exitcode == 0 for p in procs)


def _one_file_worker(rank, world, port, files, q):
    """one building for two ranks (rank 1 owns none and still joins the gathers with an empty contribution), and a
    capacity (max_det 3) below both the detections (5) and the ground truth (22 boxes) of the building"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from detection_3d_amd import engine
    from detection_3d_amd.config import get_cfg
    cfg = get_cfg("4c_Fpn432")
    model = _StandIn()
    res = engine.inference(model, cfg, files, None, max_det=3, voxelize_fn=_cpu_voxelize)
    if rank == 0:
        dets, gts = res
        ok = sorted(dets) == [0] and dets[0]["bbox3d"].shape == (5, 7) and gts[0]["bbox3d"].shape[0] == 22
        q.put(bool(ok))
    elif res is not None:
        q.put(False)
    dist.barrier()
    dist.destroy_process_group()


def test_inference_keeps_every_box_and_serves_a_rank_without_buildings(tmp_path):
    from detection_3d_amd.synthetic import write_scene_file
    files = [write_scene_file(str(tmp_path / "only.npz"), 3, 2000)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_one_file_worker, args=(r, 2, port, files, q)) for r in range(2)]
    [p.start() for p in procs]
    ok = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert ok and all(p.exitcode == 0 for p in procs)


def test_pack_detections_refuses_to_truncate():
    import pytest
    from detection_3d_amd.distributed import agree_capacity, pack_detections
    r = _fake_result(0)
    n = r["bbox3d"].shape[0]
    with pytest.raises(ValueError):
        pack_detections(0, r, n - 1)
    assert agree_capacity([n, 1], 2) == n and agree_capacity([], 7) == 7
    assert pack_detections(0, r, agree_capacity([n], 1)).shape == (n + 1, 9)

"""Loops around the detector: the reference's `inference` (maskrcnn_benchmark/engine/inference_3d.py:16-36,
compute_on_dataset + gather + evaluate) and `do_train` (engine/trainer_sparse3d.py:42-160), composed from this package's
pieces so that one process per GPU feeds itself:

    ScenePrefetcher (files[rank::world], pinned host buffers, copy on a side stream)
      -> d3d_voxelize -> SparseRCNN -> pack_detections / gather_detections (one all_gather of padded tensors)
      -> eval_detection_suncg on rank 0

There is no collective on the inference data path; training adds DistributedDataParallel's bucketed gradient all-reduce
(RCCL over xGMI) and the small loss `reduce` for logging."""
import time

import torch
import torch.distributed as dist

from . import training as T
from .distributed import agree_capacity, gather_detections, pack_detections
from .scene_io import ScenePrefetcher


def _hip_voxelize(pcl, cfg):
    from .voxelize import voxelize          # d3d_voxelize: needs the GPU library
    return voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def inference(model, cfg, files, device, depth=2, max_det=None, voxelize_fn=_hip_voxelize):
    """Detections of every building in `files`, sharded over the ranks of the default process group.
    -> on rank 0: ({file index: detections dict}, {file index: targets dict of the building in the detector's frame});
    None on the other ranks.  Targets travel with the detections so that rank 0 can evaluate without re-reading files."""
    rank, world = _rank_world()
    max_det = max_det or int(cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG) * max(1, len(cfg.MODEL.SEPARATE_CLASSES_ID) + 1)
    pre = ScenePrefetcher(files, cfg.INPUT.CLASSES, cfg.SPARSE3D.VOXEL_SCALE, device=device, rank=rank, world=world,
                          depth=depth)
    was_training = model.training
    model.eval()
    results, truths = [], []
    with torch.no_grad():
        for i, (pcl, tg, _path) in enumerate(pre):
            coords, feats = voxelize_fn(pcl, cfg)
            results.append((rank + i * world, model([coords, feats])))
            truths.append({"bbox3d": tg["bbox3d"], "labels": tg["labels"],
                           "scores": torch.ones(tg["bbox3d"].shape[0], device=tg["bbox3d"].device)})
    model.train(was_training)
    # capacities from the data (one MAX all-reduce each): a building with more ground-truth boxes -- or, with ties at the
    # cut, more detections -- than DETECTIONS_PER_IMG x groups keeps all of them
    cap_det = agree_capacity([r["bbox3d"].shape[0] for _, r in results], max_det, device)
    cap_gt = agree_capacity([t["bbox3d"].shape[0] for t in truths], 1, device)
    local = [pack_detections(sid, r, cap_det) for sid, r in results]
    local_gt = [pack_detections(sid, t, cap_gt) for (sid, _), t in zip(results, truths)]
    n = len(files)
    # a rank without buildings still takes part in the gathers, with an empty contribution on ITS device
    dets = gather_detections(local, n, cap_det, device=device)
    gts = gather_detections(local_gt, n, cap_gt, device=device)
    if rank != 0:
        return None
    return dets, {k: {"bbox3d": v["bbox3d"], "labels": v["labels"]} for k, v in gts.items()}


def evaluate(cfg, dets, gts):
    """data3d/evaluation/suncg/suncg_eval.py:714-966 on gathered results (rank 0): AP (VOC-07), AIoU, recall tables."""
    from .evaluation import eval_detection_suncg

    def host(d):
        return {k: v.detach().cpu().numpy() for k, v in d.items()}

    ids = sorted(dets.keys())
    return eval_detection_suncg([host(dets[i]) for i in ids], [host(gts[i]) for i in ids], cfg)


def train(model, cfg, files, device, steps, local_rank=None, log_every=0, depth=2, voxelize_fn=_hip_voxelize):
    """`steps` iterations of data-parallel training over `files[rank::world]` (cycled): one building per rank and step
    (IMS_PER_BATCH 1 per GPU).  `model` must already sit on `device`; it is wrapped in DistributedDataParallel when a
    process group with more than one rank exists.  -> dict(buildings_per_s, ms_per_step, last reduced losses)."""
    rank, world = _rank_world()
    model.train()
    opt = T.make_optimizer(cfg, model)
    ddp = T.wrap_ddp(model, local_rank) if world > 1 else model
    if world == 1:
        T.freeze_unused(model)
    sched = T.make_lr_scheduler(cfg, opt, examples_per_epoch=max(len(files), 1))
    it, t0, reduced = 0, None, {}
    while it < steps:
        pre = ScenePrefetcher(files, cfg.INPUT.CLASSES, cfg.SPARSE3D.VOXEL_SCALE, device=device, rank=rank,
                              world=world, depth=depth)
        if len(pre) == 0:
            raise ValueError(f"rank {rank} of {world} has no building: {len(files)} files")
        for pcl, tg, _path in pre:
            if it == 1:                      # the first iteration pays allocations and the bucket build
                if device is not None:
                    torch.cuda.synchronize(device)
                t0 = time.perf_counter()
            coords, feats = voxelize_fn(pcl, cfg)
            _, reduced = T.train_step(ddp, opt, sched, [coords, feats], tg)
            it += 1
            if log_every and rank == 0 and it % log_every == 0:
                print(f"iter {it}: " + "  ".join(f"{k} {float(v):.4f}" for k, v in sorted(reduced.items())), flush=True)
            if it >= steps:
                break
    if device is not None:
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - (t0 if t0 is not None else time.perf_counter())
    timed = max(steps - 1, 0)
    t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    return {"buildings_per_s": (world * timed / dt) if dt > 0 and timed else None,
            "ms_per_step": (1e3 * dt / timed) if timed else None, "steps_timed": timed, "world": world,
            "losses": {k: float(v.detach()) for k, v in reduced.items()}}

"""Data-parallel training of the detector, one process per GPU (BASELINE.json configs[3]: configs/3G6c fpn4321 bs=1 x N,
RCCL gradient all-reduce over xGMI; tools/train_net_sparse3d.py:52-57,170-177 + engine/trainer_sparse3d.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/train_ddp.py --config 3G6c_Fpn4321 --steps 20 [--data DIR | --scenes 8 --points 500000]

Every rank reads its own buildings (files[rank::world]) through scene_io.ScenePrefetcher, runs forward + backward (DDP
all-reduces ~128 MB of fp32 gradients bucket by bucket during the backward pass; the never-used top-down modules are
frozen so that no per-step graph search is needed), SGD step, LR schedule; the 4-12 loss scalars are reduced to rank 0
for logging.  Rank 0 prints one JSON line: buildings/s over all ranks (max-over-ranks time), ms per step, last losses.
The process group is created BEFORE anything touches the GPU."""
import argparse
import json
import os
import sys
import tempfile

WORLD = max(1, int(os.environ.get("WORLD_SIZE", "1")))
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 1) // WORLD))))   # ranks share the host

import torch                               # noqa: E402
import torch.distributed as dist           # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3G6c_Fpn4321")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--data", default=None, help="directory of scene files (.pth / .npz); default: synthetic scenes")
    ap.add_argument("--scenes", type=int, default=0, help="synthetic scenes to write (default: 2 per rank)")
    ap.add_argument("--points", type=int, default=500_000)
    ap.add_argument("--log-every", type=int, default=0)
    ap.add_argument("--verify", action="store_true",
                    help="after the steps: compare the weights of all ranks and gather the detections of every scene")
    args = ap.parse_args()
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend=os.environ.get("D3D_DIST_BACKEND", "nccl"), init_method="env://", rank=rank,
                            world_size=WORLD)     # nccl = RCCL; gloo: rehearsals of N ranks on one GPU
    local_rank %= max(1, torch.cuda.device_count())     # (several ranks on one GPU only in gloo rehearsals)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))

    from detection_3d_amd import _lib, engine
    _lib.lib()
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.synthetic import write_scene_file

    cfg = get_cfg(args.config)
    if args.data:
        files = sorted(os.path.join(args.data, f) for f in os.listdir(args.data) if f.endswith((".pth", ".npz")))
    else:
        n = args.scenes or 2 * WORLD
        tmp = os.path.join(tempfile.gettempdir(), f"d3d_train_scenes_{os.environ.get('MASTER_PORT', '0')}")
        files = [os.path.join(tmp, f"scene_{i}.npz") for i in range(n)]
        if rank == 0:
            os.makedirs(tmp, exist_ok=True)
            for i, f in enumerate(files):
                if not os.path.exists(f):
                    write_scene_file(f, i, args.points, cfg.INPUT.CLASSES)
        dist.barrier()
    torch.manual_seed(0)                  # same initial weights on every rank (DDP also broadcasts rank 0's)
    model = build_detection_model(cfg).to(dev)
    out = engine.train(model, cfg, files, dev, args.steps, local_rank=local_rank, log_every=args.log_every)
    if args.verify:
        # (1) the averaged-gradient steps leave every rank with the same weights (fingerprint: sum and sum of squares of
        # every parameter in fp64); (2) the sharded inference loop returns every scene's detections on rank 0
        with torch.no_grad():
            fp = torch.stack([torch.stack([p.detach().double().sum(), (p.detach().double() ** 2).sum()])
                              for p in model.parameters()]).reshape(-1)
        if dist.get_backend() == "gloo":
            fp = fp.cpu()
        fps = [torch.empty_like(fp) for _ in range(WORLD)]
        dist.all_gather(fps, fp)
        out["weights_equal"] = bool(all(torch.equal(fps[0], f) for f in fps))
        res = engine.inference(model, cfg, files, dev)
        if rank == 0:
            dets, gts = res
            out["scenes_gathered"] = sorted(int(k) for k in dets)
            out["detections_per_scene"] = [int(dets[k]["bbox3d"].shape[0]) for k in sorted(dets)]
            out["gt_per_scene"] = [int(gts[k]["bbox3d"].shape[0]) for k in sorted(gts)]
    if rank == 0:
        out.update(config=args.config, n_gpus=WORLD, points_per_building=args.points if not args.data else None,
                   unit="buildings/s", metric="training buildings/sec (forward + backward + SGD, DDP)")
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

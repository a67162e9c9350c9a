"""Headline benchmark: buildings/s of full-detector inference (configs/4c fpn432, bs=1 per GPU, fp32) on
synthetic SYNBIM-shaped scenes (500 k points), N GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one building through the whole hot path with the point cloud already resident in HBM:
voxelize -> voxel hash-scatter -> rulebooks -> sparse FPN -> RPN -> rotated NMS -> rotated 3-D RoIAlign ->
box head -> per-class rotated NMS.  Buildings are independent: rank r processes scenes r, r+N, ...
(weak scaling, no data-path collective); value = buildings of all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP events on the launch stream inside
the timed region) and, at N=1, `cpu_baseline` (the CPU oracle port on a bounded sample).
"""
import argparse
import hashlib
import json
import os
import sys
import time

# host threads per rank: decided before any OpenMP runtime (torch's, the CPU oracle's) is loaded.  One rank may use the
# CPU share of a one-GPU box (16); N ranks split the cores so that they do not fight for them.
CPU_BASELINE_THREADS = 16
_WORLD = max(1, int(os.environ.get("WORLD_SIZE", "1")))
_HOST_THREADS = max(1, min(CPU_BASELINE_THREADS, (os.cpu_count() or 1) // _WORLD))
os.environ["OMP_NUM_THREADS"] = str(_HOST_THREADS)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=500_000)
    ap.add_argument("--scenes", type=int, default=4, help="distinct synthetic scenes cycled per rank")
    ap.add_argument("--profile-every", type=int, default=int(os.environ.get("D3D_BENCH_SAMPLE", "5")),
                    help="HIP events around the dominant convolution family's launches in every n-th timed pass")
    ap.add_argument("--in-flight", type=int, default=2, help="buildings in flight of the extra `pipelined` region (1: skip it)")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra `bf16_bs4` region (BASELINE.json configs[4])")
    ap.add_argument("--bf16-points", type=int, default=1_000_000)
    ap.add_argument("--bf16-batch", type=int, default=4)
    ap.add_argument("--bf16-steps", type=int, default=5)
    ap.add_argument("--bf16-only", action="store_true",
                    help="run only the `bf16_bs4` region and print its object (profiling runs of that region alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-points", type=int, default=500_000)
    return ap.parse_args()


def pmc_traffic(template, section=None, source="conv.hip"):
    """-> (HBM-side bytes per launch of `template` or None, provenance string).  `section`: sub-object of
    profiles/pmc_current.json holding the passes of another kernel file (`source`, e.g. the bf16 region's).  The PMC counters cannot be read from
    inside the benchmark process, so the figure is replayed from the committed passes named by profiles/pmc_current.json
    (`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of this script, summed per kernel name by
    scripts/pmc_summary.py, in KB) -- but only while csrc/conv.hip still has the SHA-256 it had when they were taken;
    otherwise traffic is null and the reason is reported.  gfx950 correction of MI355X_MICROARCH.md (HBM section):
    FETCH_SIZE counts wide coalesced reads at half their bytes."""
    import csv
    meta_path = os.path.join(ROOT, "profiles", "pmc_current.json")
    if not os.path.exists(meta_path):
        return None, "no profiles/pmc_current.json"
    with open(meta_path) as f:
        meta = json.load(f)
    if section is not None:
        meta = meta.get(section)
        if not meta:
            return None, f"no '{section}' passes in profiles/pmc_current.json"
    with open(os.path.join(ROOT, "detection_3d_amd", "csrc", source), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    want = meta.get("sha256", meta.get("conv_hip_sha256"))
    if sha != want:
        return None, f"stale: {meta.get('fetch_csv')} was taken for another csrc/{source} ({str(want)[:12]})"
    tot = 0.0
    for name, factor in ((meta["fetch_csv"], 2.0), (meta["write_csv"], 1.0)):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None, f"missing profiles/{name}"
        hit = None
        with open(path) as f:
            for row in csv.reader(f):
                if row and template in row[0]:
                    hit = float(row[3]) * 1024.0 * factor
                    break
        if hit is None:
            return None, f"kernel {template} not in profiles/{name}"
        tot += hit
    return tot, (f"bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE of profiles/{meta['fetch_csv']} / {meta['write_csv']} "
                 f"(separate --pmc passes, taken at {meta.get('taken_at', '?')}, {source} {sha[:12]})")


def kernel_name(key):
    """(Cin, Cout) -> the k_conv instantiation launch_conv picks (detection_3d_amd/csrc/conv.hip), as rocprofv3 prints it"""
    cin, cout = key
    cp = next(c for c in (16, 32, 64, 128, 256) if cin <= c)
    ct, nct = min(cp, 128), max(cp // 128, 1)
    vec = cin == cp
    late = vec and ct >= 32        # launch_t: the late-gather form for 16-byte row pieces and Cin tiles of >= 32 channels
    return (f"d3d::k_conv<{ct}, {nct}, {cout}, 1, {4 if cout == 32 else 1}, {'true' if vec else 'false'}, "
            f"{'true' if late else 'false'}> (Cin={cin}, Cout={cout}, all filter volumes)")


CPU_BASELINE_BUILDINGS = 3   # bounded sample: ~13 s of wall time on 16 threads


def cpu_baseline(cfg, state_dict, n_points):
    """The CPU oracle port (oracle/detector_port.py) timed on this host's cores on a bounded sample:
    one scene of `n_points` points over the same 25 x 19 x 2.7 m footprint (same number of pyramid
    levels and head work as the 500 k-point workload, fewer active voxels)."""
    threads = _HOST_THREADS                           # OMP_NUM_THREADS was set before any OpenMP runtime loaded
    torch.set_num_threads(threads)
    import oracle
    from oracle.detector_port import OracleDetector
    from detection_3d_amd.synthetic import make_scene
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    det = OracleDetector(sd, cfg)
    scenes = [make_scene(1000 + i, n_points) for i in range(CPU_BASELINE_BUILDINGS)]
    t0 = time.time()
    for pcl in scenes:
        coords, feats = oracle.voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        det(coords, feats)
    dt = time.time() - t0
    return {"value": len(scenes) / dt, "unit": "buildings/s", "cores": threads,
            "kind": "port",
            "sample": f"{len(scenes)} synthetic buildings of {n_points} points each (the GPU workload's scene "
                      f"generator, other seeds), full detector, {dt:.1f} s wall on {threads} OpenMP threads"}


def bf16_region(args, cfg, model, dev, rank, world, barrier):
    """configs[4]: `--bf16-batch` examples of `--bf16-points` points per step, backbone in bf16 storage (fp32 accumulate),
    tail in fp32; a step = voxelize each example + one batched forward pass.  Reports buildings/s and, for the bf16
    convolution kernel with the largest summed time, its compulsory bytes per second
    against the HBM peak -- at bf16 every sparse convolution is bandwidth-bound."""
    from detection_3d_amd.sparseconvnet import SCN
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize
    B, n_pts, steps = args.bf16_batch, args.bf16_points, args.bf16_steps
    clouds = []
    prof = SCN.ConvProfiler()

    def step(learn=False):
        prof.start_scene("bf16", learn)
        cs, fs = [], []
        for b, pcl in enumerate(clouds):
            c, f = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
            cs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64, device=dev)], 1))
            fs.append(f)
        return model([torch.cat(cs), torch.cat(fs), B])

    # every rank reaches the same barriers / reductions whether or not its own steps succeed (a rank that left through an
    # exception while the others wait in a collective would hang the whole job)
    failed, res = None, []
    model.backbone.compute_dtype = torch.bfloat16
    SCN.set_profiler(prof)
    try:
        clouds += [torch.from_numpy(make_scene(500 + rank * B + b, n_pts, (35.0, 27.0, 2.7))).to(dev) for b in range(B)]
        step(learn=True)
        res = step()
        torch.cuda.synchronize()
    except Exception as e:                             # noqa: BLE001
        failed = f"{type(e).__name__}: {e}"
    prof.records = []
    barrier()
    t0 = time.perf_counter()
    if failed is None:
        try:
            for _ in range(steps):
                step()
        except Exception as e:                         # noqa: BLE001
            failed = f"{type(e).__name__}: {e}"
    barrier()
    dt = time.perf_counter() - t0
    SCN.set_profiler(None)
    model.backbone.compute_dtype = torch.float32
    t = torch.tensor([dt, 0.0 if failed is None else 1.0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0].item())
    if float(t[1].item()) > 0:
        return {"value": None, "error": (failed or "another rank failed")[:300]}
    summ = prof.summary()
    out = {"value": round(world * B * steps / dt, 3), "unit": "buildings/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "dtype": "bf16 storage, fp32 accumulate (backbone); fp32 tail", "batch_per_gpu": B, "points_per_building": n_pts,
           "steps": steps, "detections": [int(r["bbox3d"].shape[0]) for r in res]}
    if summ:
        key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
        sec = d["ms"] * 1e-3
        traffic, traffic_src = pmc_traffic(f"k_conv_bf16<{key[0] if key[0] <= 128 else 128}, {max(key[0] // 128, 1)}, {key[1]},",
                                           section="bf16", source="conv_bf16.hip")
        out["roofline"] = {"bound": "hbm", "achieved": round(d["bytes"] / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(d["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_unit": traffic_src,
                           "kernel": f"d3d::k_conv_bf16 (Cin={key[0]}, Cout={key[1]})",
                           "launches_per_step": d["calls"] / steps, "avg_launch_us": round(d["ms"] / d["calls"] * 1e3, 1),
                           "tflops": round(d["flops"] / sec / 1e12, 1),
                           "all_sparse_conv_ms_per_step": round(sum(v["ms"] for v in summ.values()) / steps, 3)}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("D3D_DIST_BACKEND", "nccl"), init_method="env://")   # nccl = RCCL; gloo: rehearsals of N ranks on one GPU
    local_rank %= max(1, torch.cuda.device_count())     # (several ranks on one GPU only in gloo rehearsals)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from detection_3d_amd import _lib
    _lib.lib()   # fail loudly without the HIP library
    from detection_3d_amd.config import get_cfg
    from detection_3d_amd.detector import build_detection_model
    from detection_3d_amd.sparseconvnet import SCN
    from detection_3d_amd.synthetic import make_scene
    from detection_3d_amd.voxelize import voxelize

    cfg = get_cfg("4c_Fpn432")
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(dev).eval()
    # rank r owns scenes r, r + world, ... (seeds); resident in HBM before the timed region
    scenes = [torch.from_numpy(make_scene(rank + world * i, args.points)).to(dev) for i in range(args.scenes)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.bf16_only:
        out = bf16_region(args, cfg, model, dev, rank, world, barrier)
        if rank == 0:
            print(json.dumps({"bf16_bs4": out}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    prof = SCN.ConvProfiler()

    def step(i, learn=False):
        pcl = scenes[i % len(scenes)]
        prof.start_scene(i % len(scenes), learn)
        coords, feats = voxelize(pcl, cfg.SPARSE3D.VOXEL_SCALE, cfg.SPARSE3D.VOXEL_FULL_SCALE)
        return model([coords, feats])

    # set-up pass (untimed, not a warm-up step): learn the rule count of every conv call of every
    # distinct scene, so that the timed region only records events (no extra synchronisation)
    SCN.set_profiler(prof)
    for i in range(len(scenes)):
        step(i, learn=True)
    # warm-up steps: every convolution family is timed once more to find the dominant one (by summed time); the
    # timed region then records events for that family only (2 event records per launch, not per convolution)
    n_det = 0
    for i in range(args.warmup):
        n_det = step(i)["bbox3d"].shape[0]
    torch.cuda.synchronize()
    warm = prof.summary()
    families = [{"kernel": kernel_name(k), "ms_per_step": round(v["ms"] / max(args.warmup, 1), 3),
                 "launches_per_step": v["calls"] / max(args.warmup, 1),
                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                for k, v in sorted(warm.items(), key=lambda kv: -kv[1]["ms"])[:6]] if warm else []
    conv_ms_warm = sum(v["ms"] for v in warm.values()) / max(args.warmup, 1) if warm else None
    if warm:
        prof.focus = {max(warm.items(), key=lambda kv: kv[1]["ms"])[0]}
    prof.records = []
    # the timed region records events around the dominant family's launches of every `--profile-every`-th pass (two
    # records per launch, ~0.1 ms per pass at 20 launches): the roofline's average is over those passes' launches
    prof.sample_every, prof.sampled, prof._started = max(1, args.profile_every), 0, 0
    if warm:    # the events of the timed region's launches, created ahead of it
        prof.reserve(2 * (warm[next(iter(prof.focus))]["calls"] // max(args.warmup, 1) + 1) * args.steps)
    barrier()
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(args.steps):
        step(args.warmup + i)
        marks.append(time.perf_counter())      # (a pass ends with its detections' count on the host: these are pass times)
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(1e3 * (b - a) for a, b in zip(marks[:-1], marks[1:]))
    SCN.set_profiler(None)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # second region (reported beside `value`, never as it): the same K buildings with `--in-flight` of them going at
    # once on their own host thread + HIP stream (detection_3d_amd/serving.py), still one bs=1 pass per building
    piped = None
    if args.in_flight > 1:
        from detection_3d_amd.serving import BuildingPipeline
        failed = None
        try:
            pipe = BuildingPipeline(model, cfg, in_flight=args.in_flight, device=dev)
            order = [scenes[(args.warmup + i) % len(scenes)] for i in range(args.steps)]
            pipe.map(order[:2 * args.in_flight])      # warm-up of the worker streams (arenas, scratch)
        except Exception as e:                         # noqa: BLE001 - the extra region must never cost the main line
            failed = f"{type(e).__name__}: {e}"
        barrier()
        t0 = time.perf_counter()
        if failed is None:
            try:
                pipe.map(order)
            except Exception as e:                     # noqa: BLE001
                failed = f"{type(e).__name__}: {e}"
        barrier()
        tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        if failed is None:
            piped = {"in_flight": args.in_flight, "value": round(world * args.steps / float(tp.item()), 3),
                     "unit": "buildings/s", "ms_per_step": round(1e3 * float(tp.item()) / args.steps, 3),
                     "note": "same K buildings, bs=1 passes overlapped on separate HIP streams; results bit-identical"}
        else:
            piped = {"in_flight": args.in_flight, "value": None, "error": failed[:300]}

    # third region (extra field, never `value`): BASELINE.json configs[4] -- bf16 sparse conv, ~1 M-point scenes
    # (35 x 27 x 2.7 m), bs = 4 examples per step per GPU, same detector and weights
    bf16 = None
    if not args.no_bf16:
        try:
            bf16 = bf16_region(args, cfg, model, dev, rank, world, barrier)
        except Exception as e:                         # noqa: BLE001 - the extra region must never cost the main line
            bf16 = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        summ = prof.summary()
        # dominant sparse-conv kernel = the k_conv template instantiation (Cin, Cout) with the largest summed time;
        # all its launches (every filter volume / conv kind) are timed, so that the average launch duration is the
        # one `rocprofv3 --kernel-trace --stats` reports for that kernel name (profiles/r01_bench_kernel_stats.csv)
        if summ:
            key, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            per_launch_ms = d["ms"] / d["calls"]
            tflops = d["flops"] / d["calls"] / (per_launch_ms * 1e-3) / 1e12
            gbs = d["bytes"] / d["calls"] / (per_launch_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(kernel_name(key).split(" (")[0].replace("d3d::", ""))
            # the same launches split by size: the family's average mixes two large launches per building that run near
            # the kernel's ceiling with a dozen few-row ones that are bound by their launch -> index -> row -> MFMA chain
            buckets = []
            for lo, hi, tag in ((10e9, float("inf"), ">= 10 GFLOP"), (1e9, 10e9, "1-10 GFLOP"), (0.0, 1e9, "< 1 GFLOP")):
                sel = [(f, s_.elapsed_time(e_)) for k_, f, _b, s_, e_ in prof.records if k_ == key and lo <= f < hi]
                if sel:
                    fl, ms = sum(f for f, _ in sel), sum(m for _, m in sel)
                    buckets.append({"launches_of": tag, "launches_per_step": len(sel) / max(prof.sampled, 1),
                                    "share_of_time": round(ms / d["ms"], 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 1),
                                    "frac": round(fl / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4)})
            roof = {"bound": "mfma", "achieved": round(tflops, 3), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / FP32_MATRIX_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": traffic_src,
                    "kernel": kernel_name(key),
                    "launches_per_step": d["calls"] / max(prof.sampled, 1), "avg_launch_us": round(per_launch_ms * 1e3, 1),
                    "timed_passes": prof.sampled,
                    "algorithmic_gflop_per_launch": round(d["flops"] / d["calls"] / 1e9, 3),
                    "compulsory_GBps": round(gbs, 1), "compulsory_frac_of_hbm": round(gbs / HBM_PEAK_GBS, 4),
                    "all_sparse_conv_ms_per_step_warmup": None if conv_ms_warm is None else round(conv_ms_warm, 3),
                    "by_launch_size": buckets, "kernels_warmup": families}
        else:       # nothing was timed (no steps): the headline line is still printed
            roof = {"bound": "mfma", "achieved": None, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "traffic_unit": "no sparse convolution was timed"}
        out = {
            "metric": "buildings/sec inference, 4c_fpn432", "value": round(world * args.steps / dt_max, 3),
            "unit": "buildings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3),
            "ms_per_step_spread": {"min": round(per_step[0], 3), "median": round(per_step[len(per_step) // 2], 3),
                                   "max": round(per_step[-1], 3), "note": "rank 0, host clock per pass"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs/4c fpn432 bs=1 fp32 full-detector inference, synthetic SYNBIM-shaped "
                                   f"scene of {args.points} points (25x19x2.7 m), random-init weights",
                       "points_per_building": args.points, "buildings_per_step_per_gpu": 1,
                       "detections_last_warmup": int(n_det), "sharding": "one building per GPU, no collective"},
            "roofline": roof,
        }
        if piped:
            out["pipelined"] = piped
        if bf16:
            out["bf16_bs4"] = bf16
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model.state_dict(), args.cpu_baseline_points)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

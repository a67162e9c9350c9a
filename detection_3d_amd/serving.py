"""Several buildings in flight on one GPU (inference throughput).

One building at batch size 1 leaves the GPU under-occupied for a good part of its pass: building the grids and
rulebooks is a chain of small kernels with one host read-back per strided grid, the coarse FPN levels launch
convolutions over a few hundred rows, the detector tail reads two counts back and sweeps NMS with a single wave.
`BuildingPipeline` overlaps these phases of *consecutive* buildings, every building still an independent bs=1 pass:

    stage 1  geometry  (voxelize, input layer, all grids + rulebooks; read-backs)     thread "geometry"
    stage 2  features  (sparse-conv FPN; enqueue only, never waits for the GPU)       the calling thread
    stage 3  tail      (RPN decode + NMS, RoIAlign, box head, per-class NMS)          thread "tail"

Building i owns slot i % in_flight: a HIP stream for its geometry and tail stages (chains of small dependent kernels
whose latency sets the pipeline's rate; normal queue priority -- high priority, round 2's choice, helped on some boxes and
held the other building's convolutions back on others) and a stream for its feature pass, chained by events; its own metadata arena and scratch
(SCN._scratch_key), so no two buildings share mutable state; a semaphore bounds the buildings in flight.  While one
building's large convolutions run, the next one's geometry kernels and the previous one's tail fill the idle CUs and
hide their read-back latencies.  Results are
bit-identical to the serial loop (tests/test_detector_gpu.py); the reference has no counterpart (its test loop,
maskrcnn_benchmark/engine/inference.py:17-40 `compute_on_dataset`, is serial).
"""
import os
import queue
import threading

import torch

from .voxelize import voxelize


class BuildingPipeline(object):
    def __init__(self, model, cfg, in_flight=2, device=None):
        self.model, self.cfg = model, cfg
        self.device = device if device is not None else next(model.parameters()).device
        self.in_flight = max(1, int(in_flight))
        # streams live as long as the pipeline: metadata arenas and scratch buffers are recycled per stream
        # (normal queue priority: as high-priority streams the geometry / tail stages held back the other building's feature
        #  pass on some boxes -- 203-208 against 204-216 buildings/s in pairs of runs; D3D_PIPE_PRIORITY=-1 restores it)
        prio = int(os.environ.get("D3D_PIPE_PRIORITY", "0"))
        self.hi = [torch.cuda.Stream(device=self.device, priority=prio) for _ in range(self.in_flight)]
        self.lo = [torch.cuda.Stream(device=self.device) for _ in range(self.in_flight)]

    def map(self, clouds):
        """clouds: list of float32 [N, F] point clouds resident on the device -> list of detection dicts, in order.
        The caller's stream is made to wait for the results (no device-wide synchronisation)."""
        clouds = list(clouds)
        n = len(clouds)
        if n == 0:
            return []
        s3d = self.cfg.SPARSE3D
        caller = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(caller)
        for st in self.hi:
            st.wait_event(ready)
        results, errors = [None] * n, []
        slots = threading.Semaphore(self.in_flight)
        q_feat, q_tail = queue.Queue(), queue.Queue()
        stop = threading.Event()

        def slot(i):
            return self.hi[i % self.in_flight], self.lo[i % self.in_flight]

        def guarded(fn, downstream):
            def run():
                try:
                    torch.cuda.set_device(self.device)
                    with torch.no_grad():
                        fn()
                except BaseException as e:      # noqa: BLE001 - re-raised in the caller
                    errors.append(e)
                    stop.set()
                    for _ in range(self.in_flight + 1):
                        slots.release()
                finally:
                    if downstream is not None:
                        downstream.put(None)
            return run

        def geometry():
            for i in range(n):
                slots.acquire()
                if stop.is_set():
                    return
                hi, lo = slot(i)
                with torch.cuda.stream(hi):
                    hi.wait_stream(lo)      # the slot's previous building has left its arena and allocator blocks
                    coords, feats = voxelize(clouds[i], s3d.VOXEL_SCALE, s3d.VOXEL_FULL_SCALE)
                    net = self.model.stage_geometry([coords, feats])
                    lo.wait_stream(hi)
                q_feat.put((i, net))

        def features():
            while True:
                item = q_feat.get()
                if item is None or stop.is_set():
                    return
                i, net = item
                hi, lo = slot(i)
                with torch.cuda.stream(lo):
                    feats = self.model.stage_features(net)
                    hi.wait_stream(lo)
                q_tail.put((i, feats))
                del net, feats, item

        def tail():
            while True:
                item = q_tail.get()
                if item is None or stop.is_set():
                    return
                i, feats = item
                with torch.cuda.stream(slot(i)[0]):
                    results[i] = self.model.stage_tail(feats)
                del feats, item
                slots.release()

        threads = [threading.Thread(target=guarded(geometry, q_feat), name="d3d-geometry"),
                   threading.Thread(target=guarded(tail, None), name="d3d-tail")]
        for th in threads:
            th.start()
        guarded(features, q_tail)()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        for st in self.hi:
            caller.wait_stream(st)
        for r in results:
            for v in r.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(caller)
        return results

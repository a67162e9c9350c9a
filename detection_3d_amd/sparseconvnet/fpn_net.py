"""Sparse 3-D FPN backbone: same module tree / parameter names as
SparseConvNet/sparseconvnet/fpn_net.py:12-137 (so `backbone.*` checkpoint keys load unchanged),
same outputs as its forward_fpn (:168-265), executed by the HIP ops of libd3d_hip.so.

Execution switches that do not change any returned tensor (TWO_LANE below is a third):
  * fuse_adds     -- residual / lateral additions run in the epilogue of the producing
                     convolution instead of a separate AddTable / add_feature_planes pass;
  * skip_unused   -- the top-down levels whose outputs nothing consumes for the configured
                     fpn_scales_from_top / roi_scales_from_top (m_ups[5..7] for fpn432) are not
                     computed; the reference computes and discards them (fpn_net.py:186-196).
"""
import numpy as np
import os

import torch
import torch.nn as nn

from . import modules as scn
from ..timeline import mark as _tmark

TWO_LANE = True     # grid chain of the coarser levels on a side stream while the finer ones convolve
PLAN_LANE = os.environ.get("D3D_PLAN_LANE", "0") != "0"   # ... and each level's submanifold / deconvolution rulebooks on a third
                    # (measured: 6.2-6.6 ms per building against 6.1-6.4 without -- the kernels it overlaps slow each other down)
ASYNC_GEOMETRY = os.environ.get("D3D_ASYNC_GEOMETRY", "1") != "0"   # the grid chain is run by a thread of the library
                    # (d3d_geometry_async_*): its count read-backs no longer stop this thread from enqueueing
ASYNC_VIEWS = os.environ.get("D3D_ASYNC_VIEWS", "1") != "0"         # ... and that thread also enqueues the views (third stream)
SIDE_START = os.environ.get("D3D_SIDE_START", "scene")              # "main": side streams wait for the caller's stream at the
                    # input grid (and the point lists go to the geometry stream), as before the grid chain -- A/B runs
_GEO_STREAMS = {}   # (device, caller's stream) -> side streams
# priority of the side streams: 0 = normal (default), -1 = high (geometry ahead of the convolutions).  Measured in pairs on
# two boxes: 4.80 (high) against 4.82 ms (normal) per building on one, 5.28-5.37 against 4.86-4.95 on the other -- high
# priority buys nothing where it works and costs 8 % where the queue scheduler lets the geometry kernels hold back the
# caller's stream
_SIDE_PRIORITY = int(os.environ.get("D3D_SIDE_PRIORITY", "0"))


def _is_gpu_input(net0):
    return isinstance(net0, (list, tuple)) and len(net0) > 1 and torch.is_tensor(net0[1]) and net0[1].is_cuda


def _geometry_stream(main):
    """-> (geometry stream, its reusable per-level events, plan stream, its events) of the caller's stream"""
    key = (main.device.index, main.cuda_stream)
    st = _GEO_STREAMS.get(key)
    if st is None:
        st = _GEO_STREAMS[key] = (torch.cuda.Stream(device=main.device, priority=_SIDE_PRIORITY),
                                  [torch.cuda.Event() for _ in range(16)],
                                  torch.cuda.Stream(device=main.device, priority=_SIDE_PRIORITY),
                                  [torch.cuda.Event() for _ in range(16)])
    return st


class FPN_Net(torch.nn.Module):
    def __init__(self, full_scale, dimension, raw_elements, reps, nPlanesF, nPlaneM, residual_blocks,
                 fpn_scales_from_top, roi_scales_from_top, downsample, rpn_map_sizes,
                 rpn_3d_2d_selector, leakiness=0, voxel_scale=None, bn_momentum=0.9,
                 track_running_stats=True, fuse_adds=True, skip_unused=True):
        nn.Module.__init__(self)
        self.bn_momentum = bn_momentum
        self.track_running_stats = track_running_stats
        self.dimension = dimension
        self.down_kernels, self.down_strides = downsample[0], downsample[1]
        self.fpn_scales_from_top = list(fpn_scales_from_top)
        self.roi_scales_from_top = list(roi_scales_from_top)
        self.residual_blocks = residual_blocks
        self.reps = reps
        self.fuse_adds, self.skip_unused = fuse_adds, skip_unused
        # storage type of the feature maps between the input layer and the maps handed to RPN / pooler (inference):
        # torch.bfloat16 = BASELINE.json configs[4] (bf16 rows and weights, fp32 accumulation, fp32 statistics)
        self.compute_dtype = torch.float32
        n_scales = len(nPlanesF)
        assert len(self.down_kernels) == n_scales - 1 == len(self.down_strides)
        in_channels = sum({'xyz': 3, 'color': 3, 'normal': 3}[e] for e in raw_elements)
        bn = dict(momentum=bn_momentum, track_running_stats=track_running_stats)

        self.layers_in_0 = scn.Sequential(scn.InputLayer(dimension, full_scale, mode=4))
        self.layers_in = scn.Sequential(
            scn.InputLayer(dimension, full_scale, mode=4),
            scn.SubmanifoldConvolution(dimension, in_channels, nPlanesF[0], 3, False))
        self.layers_out = scn.Sequential(scn.BatchNormReLU(nPlanesF[0], **bn), scn.OutputLayer(dimension))
        self.linear = nn.Linear(nPlanesF[0], 20)
        self.voxel_scale = voxel_scale
        self.rpn_map_sizes = np.array(rpn_map_sizes)
        self.rpn_3d_2d_selector = list(rpn_3d_2d_selector)
        self.convs_pro2d = nn.ModuleList(
            [scn.Convolution(dimension, nPlaneM, nPlaneM, [1, 1, int(z)], [1, 1, 1], False)
             for z in self.rpn_map_sizes[:, -1]])

        def block(m, a, b):
            if residual_blocks:
                assert a == b, "NetworkInNetwork branch is never taken by the detector configs"
                m.add(scn.ConcatTable()
                      .add(scn.Identity())
                      .add(scn.Sequential()
                           .add(scn.BatchNormLeakyReLU(a, leakiness=leakiness, **bn))
                           .add(scn.SubmanifoldConvolution(dimension, a, b, 3, False))
                           .add(scn.BatchNormLeakyReLU(b, leakiness=leakiness, **bn))
                           .add(scn.SubmanifoldConvolution(dimension, b, b, 3, False)))
                      ).add(scn.AddTable())
            else:
                m.add(scn.Sequential()
                      .add(scn.BatchNormLeakyReLU(a, leakiness=leakiness, **bn))
                      .add(scn.SubmanifoldConvolution(dimension, a, b, 3, False)))

        self.m_downs, self.m_shortcuts = nn.ModuleList(), nn.ModuleList()
        for k in range(n_scales):
            m = scn.Sequential()
            if k > 0:
                m.add(scn.Sequential()
                      .add(scn.BatchNormLeakyReLU(nPlanesF[k - 1], leakiness=leakiness, **bn))
                      .add(scn.Convolution(dimension, nPlanesF[k - 1], nPlanesF[k],
                                           self.down_kernels[k - 1], self.down_strides[k - 1], False)))
            for _ in range(reps):
                block(m, nPlanesF[k], nPlanesF[k])
            self.m_downs.append(m)
            self.m_shortcuts.append(scn.SubmanifoldConvolution(dimension, nPlanesF[k], nPlaneM, 1, False))

        self.m_ups, self.m_mergeds = nn.ModuleList(), nn.ModuleList()
        for k in range(n_scales - 1, 0, -1):
            self.m_ups.append(scn.Sequential()
                              .add(scn.BatchNormLeakyReLU(nPlaneM, leakiness=leakiness, **bn))
                              .add(scn.Deconvolution(dimension, nPlaneM, nPlaneM, self.down_kernels[k - 1],
                                                     self.down_strides[k - 1], False)))
            self.m_mergeds.append(scn.SubmanifoldConvolution(dimension, nPlaneM, nPlaneM, 3, False))

    # ------------------------------------------------------------------------------------
    def _to_compute(self, net):
        """input-layer output (fp32 [n, 9]) -> storage type of the backbone: bf16 rows are padded to 16 channels"""
        if self.compute_dtype == torch.float32 or net.features.dtype == self.compute_dtype:
            return net
        assert not torch.is_grad_enabled() or not net.features.requires_grad, "bf16 storage is an inference path"
        f = net.features
        width = scn.SCN.stored_planes(f.shape[1], self.compute_dtype)
        if f.is_cuda and f.dtype == torch.float32 and self.compute_dtype == torch.bfloat16:
            net.features = scn.SCN.rows_to_bf16(f, width)       # (a pad and a cast took 1 ms of a 4 x 1 M-point step)
        else:
            net.features = torch.nn.functional.pad(f, (0, width - f.shape[1])).to(self.compute_dtype)
        return net

    def _from_compute(self, maps):
        if self.compute_dtype == torch.float32:
            return maps
        return [None if t is None else scn.SparseConvNetTensor(t.features.float(), t.metadata, t.spatial_size) for t in maps]

    def forward(self, net0):
        if TWO_LANE and _is_gpu_input(net0):
            if ASYNC_GEOMETRY and not PLAN_LANE:
                return self._forward_async_geometry(net0)
            return self._forward_two_lane(net0)
        net1 = self.layers_in[1](self._to_compute(self.layers_in[0](net0)))
        return self.forward_fpn(net1)

    def _geometry_specs(self, size0, views=False):
        """The d3d_conv_prepare calls of the pyramid in the order _geometry_steps makes them, as rows of 13 ints
        (kind 1, in_size, out_size, filter, stride), and for every level the rows it needs before its convolutions may be
        enqueued: (its last grid row, its 3x3x3 view row), -1 = none.  With `views` the rulebooks that are views of a grid are listed too
        (what _geometry_steps builds with full=True): the 3x3x3 submanifold rulebook (kind 0) of every level but the
        first right behind its grid -- the row a level then waits for -- and, after all grids, the lateral 1x1x1
        rulebooks and the deconvolution views (kind 2) of the top-down path.
        -> (rows, last row per level, last row of all)"""
        n_scales = len(self.m_downs)
        n3d = len(self.fpn_scales_from_top)
        sel2d = sorted({i - n3d for i in self.rpn_3d_2d_selector if i >= n3d}) if self.skip_unused else range(n3d)
        pro2d = {n_scales - 1 - self.fpn_scales_from_top[i]: self.convs_pro2d[i] for i in sel2d}
        needed = max(self.fpn_scales_from_top + self.roi_scales_from_top) if self.skip_unused else n_scales - 1
        lowest_up = n_scales - 1 - min(n_scales - 1, needed)
        size = scn.toLongTensor(self.dimension, size0)
        three, one = [3] * self.dimension, [1] * self.dimension
        specs, last, later, sizes = [], [], [], []
        for k in range(n_scales):
            if k > 0:
                filt = scn.toLongTensor(self.dimension, self.down_kernels[k - 1])
                stride = scn.toLongTensor(self.dimension, self.down_strides[k - 1])
                out = (size - filt) // stride + 1
                # kind 3: no deconvolution / backward view of this rulebook will be asked for (inference, below the
                # finest level the top-down path reaches): its decoded table is not built
                need_dec = self.training or torch.is_grad_enabled() or k > lowest_up or not views
                specs.append([1 if need_dec else 3] + size.tolist() + out.tolist() + filt.tolist() + stride.tolist())
                if views and k > lowest_up:
                    later.append([2] + out.tolist() + size.tolist() + filt.tolist() + stride.tolist())
                size = out
            sizes.append(size)
            if k in pro2d:
                conv = pro2d[k]
                out = (size - conv.filter_size) // conv.filter_stride + 1
                need_dec = self.training or torch.is_grad_enabled()
                specs.append([1 if need_dec else 3] + size.tolist() + out.tolist() + conv.filter_size.tolist()
                             + conv.filter_stride.tolist())
            grid_row = len(specs) - 1 if (k > 0 or k in pro2d) else -1     # the level's last grid / strided rulebook
            view_row = -1
            if views:
                if k > 0:           # (level 0's is built by the caller while the point lists are sorted)
                    specs.append([0] + size.tolist() + size.tolist() + three + one)
                    view_row = len(specs) - 1
                if k >= lowest_up:
                    later.append([0] + size.tolist() + size.tolist() + one + one)
            # a level waits for both: its 3x3x3 view starts as soon as the GRID exists, before the strided rulebook
            last.append((grid_row, view_row))
        specs += later
        return specs, last, len(specs) - 1

    def _forward_async_geometry(self, net0):
        """The pass with the grid chain run by a thread of the library (d3d_geometry_async_start): every new grid costs
        a blocking read-back of its site count, and while this thread waited for one it could not enqueue the feature
        kernels of the level before -- by the end of the bottom-up path the caller's stream had caught up with its own
        launch thread.  Here the chain of all levels starts right after the input grid exists and runs at its own pace
        on the geometry stream; this thread picks a level up (count + stream dependency) when it is about to enqueue
        it.  With ASYNC_VIEWS the same thread also enqueues, on a third stream, the rulebooks that are views of a
        finished grid (3x3x3 right behind each grid; lateral and deconvolution views at the end).  Same kernels on the
        same data (bit-identical)."""
        main = torch.cuda.current_stream(net0[1].device)
        geo, pool, plan = _geometry_stream(main)[:3]
        if not ASYNC_VIEWS:
            plan = None
        plan0 = pool[-1]
        state = {}

        # both side streams start behind what the caller's stream held when the pass began (arena reuse from scene to scene
        # is ordered by that stream); they do NOT wait for what the pass itself enqueues there, see below
        scene_start = pool[-2]
        scene_start.record(main)

        def after_input_build(md, size, run_forward=None):
            # The host has just seen the input grid's site count, i.e. the grid is complete: the side streams need not
            # wait for the caller's stream, which already holds the hash probes of level 0's rulebook (~0.1 ms).
            if SIDE_START == "main":
                geo.wait_stream(main)
            else:
                geo.wait_event(scene_start)
            md.set_geometry_stream(geo.cuda_stream)
            if plan is not None and SIDE_START != "main":
                plan.wait_event(scene_start)
                with torch.cuda.stream(plan):
                    scn.SCN.InputLayer_prepare(md)          # point lists: own scratch (no lane of the arena), ~0.1 ms
                    if run_forward is not None:             # ... and the per-voxel means right behind them, beside the
                        means = run_forward(plan)           # sort of level 0's rulebook on the caller's stream
                        if means is not None and self.compute_dtype == torch.bfloat16 and means.dtype == torch.float32:
                            # ... and their bf16 rows (one launch; a pad and a cast on the caller's stream were three
                            # and sat in front of the first convolution)
                            width = scn.SCN.stored_planes(means.shape[1], self.compute_dtype)
                            state["stored"] = (means, scn.SCN.rows_to_bf16(means, width))
                    plan0.record(plan)
            else:
                with torch.cuda.stream(geo):
                    scn.SCN.InputLayer_prepare(md)
                    plan0.record(geo)
            # level 0's 3x3x3 rulebook on this stream (sort + transpose of the probed table, ~0.15 ms); the plan lane is
            # carved out of the feature lane only afterwards, and its stream continues behind this build (whose scratch
            # may reach into what becomes the plan lane)
            _tmark("input grid known (host)", -1, host=True)
            scn.SCN.SubmanifoldConvolution_prepare(size, (3,) * self.dimension, md)
            _tmark("level-0 rulebook", -1, main)
            if plan is not None:
                _tmark("point lists + input means", -1, plan)
                plan.wait_stream(main)
                md.set_plan_stream(plan.cuda_stream)
            cache = getattr(self, "_spec_cache", None)
            key = tuple(scn.SCN._size3(size)) + (plan is not None, bool(self.training), torch.is_grad_enabled())
            if cache is None or cache[0] != key:
                cache = self._spec_cache = (key,) + self._geometry_specs(size, views=plan is not None)
            state["last"], state["all"] = cache[2], cache[3]
            # the chain of strided grids starts at once on the geometry stream (one read-back for all of its levels)
            md.geometry_async_start(cache[1], geo.cuda_stream, plan.cuda_stream if plan is not None else None)
            state["md"] = md
            main.wait_event(plan0)

        # (the neighbour table of level 0's 3x3x3 rulebook is probed while the input grid's site count is read back)
        scn.SCN.set_after_input_build(after_input_build, prefetch_filter=(3,) * self.dimension)
        try:
            net = self.layers_in[0](net0)                   # input layer: grid of level 0
        finally:
            scn.SCN.set_after_input_build(None)
        md = net.metadata
        n_scales = len(self.m_downs)
        try:
            if "md" not in state:                           # (an input layer that did not go through the hook)
                after_input_build(md, net.spatial_size)

            def lane(k):
                _tmark("host enters", k, host=True)
                _tmark("main arrives", k, main)
                rows = (state["all"],) if k >= n_scales else state["last"][k]
                for idx in rows:
                    if idx >= 0 and (k < n_scales or plan is not None):
                        md.geometry_async_wait(idx, main.cuda_stream)
                _tmark("main continues", k, main)
                _tmark("host leaves", k, host=True)

            lane(0)
            stored = state.pop("stored", None)
            if stored is not None and stored[0] is net.features:
                net.features = stored[1]
            net = self.layers_in[1](self._to_compute(net))
            out = self.forward_fpn(net, prepared=True, lane=lane)
        finally:
            try:
                md.geometry_async_finish()
            finally:
                main.wait_stream(geo)
                if plan is not None:
                    main.wait_stream(plan)
                    md.set_plan_stream(None)
                md.set_geometry_stream(None)
        return out

    def _forward_two_lane(self, net0):
        """Forward pass on three HIP streams: the chain of strided grids (small dependent kernels and one count
        read-back per grid) is built on a high-priority side stream, level k+1 while the convolutions of level k run on
        the caller's stream; the rulebooks that are views of a finished grid (3x3x3 / 1x1x1 submanifold, deconvolution)
        are built on a third stream as soon as their grid exists, instead of by the first convolution that needs them,
        so neither the grid chain nor the convolutions wait behind their hash probes and sorts.  Same kernels on the
        same data as the one-stream pass (bit-identical).  Meanwhile the metadata accepts new grids on the geometry
        stream only and gives each stream its own part of the arena (d3d_meta_set_geometry_stream / _plan_stream).
        PLAN_LANE False: two streams, the views built on the caller's stream on first use (round 1)."""
        main = torch.cuda.current_stream(net0[1].device)
        geo, pool, plan, ppool = _geometry_stream(main)
        if not PLAN_LANE:
            plan = None
        plan0 = pool[-1]
        three = (3,) * self.dimension

        started = []

        def after_input_build(md, size):
            # the level-0 grid exists: its 3x3x3 rulebook (hash probes + sort, ~0.4 ms) starts at once while the
            # geometry stream sorts the input layer's point lists, which the feature pass then only has to wait for
            started.append(md)
            geo.wait_stream(main)
            md.set_geometry_stream(geo.cuda_stream)
            if plan is not None:
                plan.wait_stream(main)
                md.set_plan_stream(plan.cuda_stream)
                with torch.cuda.stream(plan):
                    scn.SCN.SubmanifoldConvolution_prepare(size, three, md)
            else:
                scn.SCN.SubmanifoldConvolution_prepare(size, three, md)
            with torch.cuda.stream(geo):
                scn.SCN.InputLayer_prepare(md)
                plan0.record(geo)
            main.wait_event(plan0)

        scn.SCN.set_after_input_build(after_input_build)
        try:
            net = self.layers_in[0](net0)                   # input layer: grid of level 0
        finally:
            scn.SCN.set_after_input_build(None)
        md = net.metadata
        if not started:                                     # (an input layer that did not go through the hook)
            geo.wait_stream(main)
            md.set_geometry_stream(geo.cuda_stream)
            if plan is not None:
                plan.wait_stream(main)
                md.set_plan_stream(plan.cuda_stream)
        steps, events, pevents = self._geometry_steps(net, False, views=plan is not None), [], []
        n_scales = len(self.m_downs)

        def lane(k):        # level k is about to be enqueued: build its grid now (level k-1 is already in the queue)
            if k >= n_scales:                               # the top-down path follows: every view has to be there
                if plan is not None:
                    main.wait_stream(plan)
                return
            _tmark("host enters", k, host=True)
            _tmark("main arrives", k, main)
            while len(events) <= k:
                i = len(events)
                ev = pool[i] if i < len(pool) - 1 else torch.cuda.Event()
                with torch.cuda.stream(geo):
                    _tmark("geo starts", i, geo)
                    views = next(steps)
                    ev.record(geo)
                    _tmark("geo done", i, geo)
                events.append(ev)
                _tmark("host has count", i, host=True)
                if plan is not None:
                    pev = ppool[i] if i < len(ppool) else torch.cuda.Event()
                    with torch.cuda.stream(plan):
                        plan.wait_event(ev)
                        _tmark("plan starts", i, plan)
                        views[0]()                          # the 3x3x3 rulebook the level's blocks need
                        pev.record(plan)
                        _tmark("plan 3x3x3 done", i, plan)
                        for v in views[1:]:                 # lateral and deconvolution views: needed on the way up
                            v()
                        _tmark("plan views done", i, plan)
                    pevents.append(pev)
            main.wait_event(events[k])
            if plan is not None:
                main.wait_event(pevents[k])
            _tmark("main continues", k, main)
            _tmark("host leaves", k, host=True)

        try:
            lane(0)
            net = self.layers_in[1](self._to_compute(net))
            out = self.forward_fpn(net, prepared=True, lane=lane)
        finally:
            main.wait_stream(geo)
            if plan is not None:
                main.wait_stream(plan)
                md.set_plan_stream(None)
            md.set_geometry_stream(None)
        return out

    def unused_modules(self):
        """Sub-modules whose parameters never receive a gradient under `skip_unused` (the reference computes some of them
        and discards the result, fpn_net.py:186-203; others -- layers_out, linear -- it only constructs): the top-down
        levels below the finest consumed map, their laterals and merges, unselected z-projections.  A data-parallel
        wrapper freezes them instead of searching the graph for them every step."""
        n_scales = len(self.m_downs)
        needed = max(self.fpn_scales_from_top + self.roi_scales_from_top) if self.skip_unused else n_scales - 1
        used_levels = min(n_scales - 1, needed)
        out = [self.layers_out, self.linear]
        out += [self.m_ups[k] for k in range(used_levels, len(self.m_ups))]
        consumed = set(self.fpn_scales_from_top) | set(self.roi_scales_from_top)
        # ups[k + 1] = m_mergeds[k](...) feeds heads only; the top-down path itself continues from the un-merged sum
        out += [self.m_mergeds[k] for k in range(len(self.m_mergeds))
                if k >= used_levels or (self.skip_unused and (k + 1) not in consumed)]
        out += [self.m_shortcuts[j] for j in range(0, n_scales - 1 - used_levels)]
        n3d = len(self.fpn_scales_from_top)
        sel2d = {i - n3d for i in self.rpn_3d_2d_selector if i >= n3d} if self.skip_unused else set(range(n3d))
        out += [self.convs_pro2d[i] for i in range(len(self.convs_pro2d)) if i not in sel2d]
        return out

    def _run_down(self, m, net):
        if not (self.fuse_adds and self.residual_blocks):
            return m(net)
        children = list(m._modules.values())
        i = 0
        while i < len(children):
            c = children[i]
            if isinstance(c, scn.ConcatTable):   # [Identity, (BN, conv, BN, conv)] followed by AddTable
                seq = c._modules['1']
                y = seq[0](net)
                y = seq[1](y)
                y = seq[2](y)
                net = seq[3](y, residual=net)     # out = conv(..) + identity branch
                i += 2
            else:
                net = c(net)
                i += 1
        return net

    def _geometry_steps(self, net, full, views=False):
        """Generator over the pyramid levels k = 0 .. n_scales-1: enqueues (on the current stream) everything level k
        needs -- the strided grid + rulebook k-1 -> k (one host read-back of the site count), the z-collapsing RPN
        projection grid of that level and, with `full`, the submanifold 3x3x3 / 1x1x1 rulebooks and the deconvolution
        view k -> k-1 (otherwise built by the first convolution that needs them) -- then yields k.  With `views` it
        yields instead the list of calls that build those rulebooks (3x3x3 first), for the caller to run on a stream of
        its choice."""
        md, n_scales = net.metadata, len(self.m_downs)
        needed = max(self.fpn_scales_from_top + self.roi_scales_from_top) if self.skip_unused else n_scales - 1
        lowest_up = n_scales - 1 - min(n_scales - 1, needed)          # finest level the top-down path reaches
        n3d = len(self.fpn_scales_from_top)
        sel2d = sorted({i - n3d for i in self.rpn_3d_2d_selector if i >= n3d}) if self.skip_unused else range(n3d)
        pro2d = {n_scales - 1 - self.fpn_scales_from_top[i]: self.convs_pro2d[i] for i in sel2d}
        sizes = [net.spatial_size]
        for k in range(n_scales):
            if k > 0:
                filt = scn.toLongTensor(self.dimension, self.down_kernels[k - 1])
                stride = scn.toLongTensor(self.dimension, self.down_strides[k - 1])
                out = (sizes[-1] - filt) // stride + 1
                scn.SCN.Convolution_prepare(sizes[-1], out, filt, stride, md)
                sizes.append(out)
            size = sizes[k]
            if k in pro2d:
                conv = pro2d[k]
                scn.SCN.Convolution_prepare(size, (size - conv.filter_size) // conv.filter_stride + 1,
                                            conv.filter_size, conv.filter_stride, md)
            todo = [lambda size=size: scn.SCN.SubmanifoldConvolution_prepare(size, (3,) * self.dimension, md)]
            if k >= lowest_up:                                                                 # lateral 1x1x1
                todo.append(lambda size=size: scn.SCN.SubmanifoldConvolution_prepare(size, (1,) * self.dimension, md))
            if k > lowest_up:
                todo.append(lambda size=size, k=k: scn.SCN.Deconvolution_prepare(
                    size, sizes[k - 1], self.down_kernels[k - 1], self.down_strides[k - 1], md))
            if full:
                for f in todo:
                    f()
            yield todo if views else k

    def prepare_geometry(self, net, full=False):
        """All strided grids / rulebooks of the pyramid, built before the first feature kernel: each new grid costs
        one host read-back of its site count, and here the stream holds only small geometry kernels when that
        happens, so the feature pass that follows is enqueued without a single synchronisation.
        full: also the submanifold and deconvolution rulebooks, which leaves the feature pass free of geometry kernels
        (serving.BuildingPipeline)."""
        for _ in self._geometry_steps(net, full):
            pass

    def stage_geometry(self, net0):
        """Stage 1 of 3 of a pipelined pass (serving.BuildingPipeline): input layer + every grid and rulebook."""
        net = self.layers_in[0](net0)
        self.prepare_geometry(net, full=True)
        return net

    def stage_features(self, net):
        """Stage 2: the feature pass over the prepared geometry (no host synchronisation)."""
        return self.forward_fpn(self.layers_in[1](self._to_compute(net)), prepared=True)

    def forward_fpn(self, net, prepared=False, lane=None):
        n_scales = len(self.m_downs)
        if not prepared:
            self.prepare_geometry(net)
        downs = []
        for k, m in enumerate(self.m_downs):
            if lane is not None:
                lane(k)
            net = self._run_down(m, net)
            downs.append(net)
        if lane is not None:
            lane(n_scales)
        _tmark("down path done")
        net = self.m_shortcuts[-1](net)
        ups = [net]
        needed = max(self.fpn_scales_from_top + self.roi_scales_from_top) if self.skip_unused else n_scales - 1
        consumed = set(self.fpn_scales_from_top) | set(self.roi_scales_from_top)
        for k in range(min(n_scales - 1, needed)):
            j = n_scales - 2 - k
            shortcut = self.m_shortcuts[j](downs[j])
            if self.fuse_adds:
                up = self.m_ups[k]
                net = up[1](up[0](net), residual=shortcut)
            else:
                net = scn.add_feature_planes([self.m_ups[k](net), shortcut])
            # a merged map that neither the RPN nor the pooler consumes is not computed (the reference computes and
            # drops it): the top-down path continues from `net`, the un-merged sum
            ups.append(self.m_mergeds[k](net) if (not self.skip_unused or (k + 1) in consumed) else None)
        _tmark("top-down done")
        rpn_maps_3d = [ups[i] for i in self.fpn_scales_from_top]
        selected_2d = {i - len(rpn_maps_3d) for i in self.rpn_3d_2d_selector if i >= len(rpn_maps_3d)}
        rpn_maps_2d = [self.convs_pro2d[i](rpn_maps_3d[i]) if (i in selected_2d or not self.skip_unused) else None
                       for i in range(len(rpn_maps_3d))]
        rpn_maps = rpn_maps_3d + rpn_maps_2d
        rpn_maps = [rpn_maps[i] for i in self.rpn_3d_2d_selector]
        roi_maps = [ups[i] for i in self.roi_scales_from_top]
        for i in range(len(rpn_maps_3d)):
            assert rpn_maps_3d[i].spatial_size.tolist() == [int(v) for v in self.rpn_map_sizes[i]]
        return self._from_compute(rpn_maps), self._from_compute(roi_maps)

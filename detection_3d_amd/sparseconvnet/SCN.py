"""Drop-in for the pybind module `sparseconvnet.SCN` of the reference
(SparseConvNet/sparseconvnet/SCN/pybind.cpp:11-235, sparseconvnet.h:9-239), for the ops the
3-D detection path uses.  Same names, argument order and ownership rules (outputs are passed in
empty and resized by the callee), but every call lands in libd3d_hip.so on the current HIP stream
-- including the rulebook construction, which the reference does on the CPU.

Forward and backward entry points of the ops the 3-D detection path uses.
"""
import ctypes
import threading
import os

import torch

from .. import _lib
from .._lib import check, ints, lib, ptr, require_gpu, stream_of

_ARENA_BYTES = int(os.environ.get("D3D_ARENA_MB", "3072")) << 20   # 2/3 geometry lane + 1/3 feature lane
_POOL = []          # recycled native metadata handles (one HBM arena each)
_POOL_MAX = int(os.environ.get("D3D_ARENA_POOL", "8"))   # idle arenas kept (3 GiB each by default) ...
_POOL_MAX_BYTES = int(os.environ.get("D3D_ARENA_POOL_GB", "32")) << 30   # ... and at most this much idle HBM
_POOL_LOCK = threading.RLock()   # re-entrant: __del__ may run inside a locked region (GC)
_SCRATCH = {}       # scratch tensors per (device, stream): buildings in flight on different streams never share one


def arena_bytes_for(n_points):
    """Arena size for a batch of `n_points` input points: the default (D3D_ARENA_MB, enough for a 500 k-point building)
    or, for larger batches, ~2.5 KiB per point (hash grids, 27-offset rulebooks in both layouts and sort temporaries of
    every scale, split 2/3 : 1/3 between the geometry and the feature lane) rounded up to whole GiB so that recycled
    arenas are found again."""
    need = int(n_points) * 2560
    if need <= _ARENA_BYTES:
        return _ARENA_BYTES
    return ((need + (1 << 30) - 1) >> 30) << 30


def _scratch_key(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (idx, _lib.raw_stream(idx))


def _scratch(device, nbytes):
    key = _scratch_key(device)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


_BN_SCRATCH = {}


def _bn_scratch(device, nbytes):
    """Scratch of the BatchNorm statistics kernels: starts with a ticket word that must be zero on entry
    (the kernel leaves it zero), so it is zero-initialised and never shared with other ops or streams."""
    key = _scratch_key(device)
    buf = _BN_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
        _BN_SCRATCH[key] = buf
    return buf


class ConvProfiler(object):
    """Optional per-launch timing of the sparse convolutions with HIP events recorded on the
    stream the kernels are launched on (bench.py's roofline leg).  Off by default.

    Rule counts live on the device and reading them costs a stream sync, so they are *learned* in an
    untimed pass (`learn=True`, one entry per conv call of a scene, keyed by `scene_key`) and looked
    up by call index while timing (`learn=False`), where only two event records per launch remain."""

    def __init__(self):
        self.learn = False
        self.focus = None   # None: time every convolution; else the set of (cin, cout) kernel keys to time
        self.scene_key = None
        self.macs = {}      # scene_key -> [macs of call 0, 1, ...]
        self._idx = 0
        self.records = []   # ((cin, cout) = one k_conv instantiation, flops, compulsory_bytes, start_event, end_event)
        self._pool = []     # timing events created ahead of the timed region (reserve)
        self.sample_every = 1   # time the convolutions of every n-th scene started with learn=False ...
        self.sampled = 0        # ... (this many so far): two event records per launch cost the pass ~2.5 us each
        self._started = 0
        self._armed = True

    def reserve(self, n_events):
        """Creates the HIP events of the next launches now: hipEventCreate inside the timed region costs host time
        the two-lane pass has none to spare of."""
        for _ in range(int(n_events)):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()         # torch creates the HIP event at its first record
            self._pool.append(ev)

    def _event(self):
        if self._pool:
            return self._pool.pop()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()             # creates the HIP event (torch does it at the first record)
        return ev

    def start_scene(self, scene_key, learn):
        self.scene_key, self.learn, self._idx = scene_key, learn, 0
        if learn:
            self.macs[scene_key] = []
            self._armed = True
        else:
            self._armed = self._started % max(1, int(self.sample_every)) == 0
            self._started += 1
            self.sampled += 1 if self._armed else 0

    def wants(self, kind, fv, cin, cout):
        if not self.learn and not self._armed:
            return False
        return self.learn or self.focus is None or any(k[:2] == (cin, cout) for k in self.focus)

    def begin(self, kind=None, fv=None, cin=None, cout=None):
        """Arms the library to bracket the k_conv launch of the convolution call that follows with two HIP events
        (d3d_conv_time_next) -> (start, stop), or None if this convolution family is not being timed."""
        if kind is not None and not self.wants(kind, fv, cin, cout):
            return None
        pair = (self._event(), self._event())
        check(lib().d3d_conv_time_next(ctypes.c_void_p(pair[0].cuda_event), ctypes.c_void_p(pair[1].cuda_event)))
        return pair

    def end(self, start, kind, fv, cin, cout, rows_in, rows_out, macs, dt=0):
        if start is None:       # not a focused family: only keep the call index aligned
            self._idx += 1
            return
        if self.learn:
            self.macs[self.scene_key].append(macs)
            return
        macs = self.macs[self.scene_key][self._idx]
        self._idx += 1
        rules = macs / max(cin * cout, 1)
        # FLOPs = 2*rules*Cin*Cout; compulsory bytes = esize*(rows_in*Cin + rows_out*Cout) + 8*rules
        esize = 2.0 if dt == BF16 else 4.0
        key = (cin, cout) if dt == F32 else (cin, cout, "bf16")
        self.records.append((key, 2.0 * macs, esize * (rows_in * cin + rows_out * cout) + 8.0 * rules,
                             start[0], start[1]))

    def summary(self):
        """{key: dict(calls, ms, flops, bytes)}; call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, nbytes, s, e in self.records:
            d = out.setdefault(key, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


PROFILER = None
# The reference's *_updateOutput return the multiply-add count (rules*Cin*Cout).  Reading the rule
# count back costs one stream synchronisation per layer, so it is only done when asked for.
COUNT_MACS = False


def set_profiler(p):
    global PROFILER
    PROFILER = p


def _size3(t):
    """Spatial size / filter / stride as a tuple of 3 ints.  LongTensors (the reference's representation) remember
    their tuple: sizes are never modified in place, and a building asks for the same ones ~200 times."""
    if type(t) is tuple and len(t) == 3:
        return t
    v = getattr(t, "_d3d_size3", None)
    if v is None:
        v = tuple(int(x) for x in (t.tolist() if hasattr(t, "tolist") else t))
        assert len(v) == 3, "only dimension 3 is built (Metadata_3)"
        if torch.is_tensor(t):
            t._d3d_size3 = v
    return v


class Metadata_3(object):
    """sparseconvnet.SCN.Metadata_3 (pybind.cpp:12-32).  Owns one HBM arena with the hash grids
    and rulebooks of one batch."""

    def __init__(self, arena_bytes=None):
        # an arena is recycled only on the stream it was used on (stream order makes the reuse safe)
        self._dev = torch.cuda.current_device()
        self._home = (self._dev, _lib.raw_stream(self._dev))
        key_bytes = arena_bytes or _ARENA_BYTES
        with _POOL_LOCK:
            hit = next((e for e in _POOL if e[0] == self._home and e[1] == key_bytes), None)
            if hit is not None:
                _POOL.remove(hit)
        if hit is not None:
            self._h, self._bytes = hit[2], hit[1]
            check(lib().d3d_meta_clear(self._h))
            return
        h = ctypes.c_void_p()
        check(lib().d3d_meta_create(ctypes.byref(h), key_bytes))
        self._h, self._bytes = h, key_bytes

    def __del__(self):
        # at interpreter shutdown module globals (and _lib's library handle) may already be gone: then the process is
        # about to release the arena anyway
        h = getattr(self, "_h", None)
        native = getattr(_lib, "_lib", None) if _lib is not None else None
        if h is None or native is None or _POOL is None or _POOL_LOCK is None:
            return
        self._h = None
        evicted = []
        with _POOL_LOCK:
            _POOL.append((self._home, self._bytes, h))
            # evict the oldest (its stream may be gone) while too many / too much idle memory is kept
            while len(_POOL) > _POOL_MAX or (len(_POOL) > 1 and sum(e[1] for e in _POOL) > _POOL_MAX_BYTES):
                evicted.append(_POOL.pop(0))
        for old in evicted:
            native.d3d_meta_destroy(old[2])

    def clear(self):
        check(lib().d3d_meta_clear(self._h))

    def set_geometry_stream(self, raw_stream):
        """d3d_meta_set_geometry_stream: rulebook / grid builds only on `raw_stream` (an int hipStream_t) from now
        on; None lifts the restriction."""
        check(lib().d3d_meta_set_geometry_stream(self._h, ctypes.c_void_p(raw_stream or 0), int(raw_stream is not None)))

    def geometry_async_start(self, specs, raw_stream, raw_view_stream=None):
        """d3d_geometry_async_start: `specs` rows of 13 ints (kind, in_size, out_size, filter, stride; kind 1 = strided grid
        on the geometry stream `raw_stream`, 0 / 2 = submanifold / deconvolution view on the plan stream
        `raw_view_stream`), built in order by a thread of the library."""
        flat = [int(v) for row in specs for v in row]
        assert len(flat) == 13 * len(specs)
        arr = (ctypes.c_int * max(1, len(flat)))(*flat)
        check(lib().d3d_geometry_async_start(self._h, arr, len(specs), ctypes.c_void_p(raw_stream),
                                             ctypes.c_void_p(raw_view_stream or 0)))

    def geometry_async_wait(self, index, raw_wait_stream):
        """-> output site count of entry `index` once it is built; `raw_wait_stream` is made to wait for it"""
        n = ctypes.c_int(0)
        check(lib().d3d_geometry_async_wait(self._h, int(index), ctypes.byref(n), ctypes.c_void_p(raw_wait_stream)))
        return n.value

    def geometry_async_finish(self):
        check(lib().d3d_geometry_async_finish(self._h))

    def set_plan_stream(self, raw_stream):
        """d3d_meta_set_plan_stream: submanifold / deconvolution rulebooks built on `raw_stream` get a lane of the arena
        of their own (a geometry stream must be set); None ends the routing."""
        check(lib().d3d_meta_set_plan_stream(self._h, ctypes.c_void_p(raw_stream or 0), int(raw_stream is not None)))

    def getNActive(self, spatial_size):
        n = ctypes.c_int(0)
        check(lib().d3d_get_n_active(self._h, ints(_size3(spatial_size)), ctypes.byref(n)))
        return n.value

    def getSpatialLocations(self, spatial_size):
        """int64 [nActive, 4] (x, y, z, batch) -- on the GPU (the reference returns a CPU tensor,
        Metadata.cpp:148-168)."""
        n = self.getNActive(spatial_size)
        out = torch.empty((n, 4), dtype=torch.int64, device=torch.device("cuda", self._dev))
        check(lib().d3d_get_spatial_locations(self._h, ints(_size3(spatial_size)), ptr(out), stream_of()))
        return out

    def anchors(self, spatial_size, base, stride, voxel_scale, out):
        """d3d_anchors: fills out [nActive*A, 7] with the anchors of every active site of this map
        (anchor_generator_sparse3d.py:86-120); base [A,7] and stride [3] are host sequences."""
        base = [float(v) for row in base for v in row]
        check(lib().d3d_anchors(self._h, ints(_size3(spatial_size)), _lib.floats(base), len(base) // 7,
                                _lib.floats(stride), float(voxel_scale), ptr(out), stream_of()))
        return out

    def arena_used(self):
        n = ctypes.c_size_t(0)
        check(lib().d3d_meta_arena_used(self._h, ctypes.byref(n)))
        return n.value

    # ---- debug exporters (canonical comparison with the oracle) ----
    def export_input_rules(self, n_points):
        n = self._in_active
        dev = torch.device("cuda", self._dev)
        off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        idx = torch.empty(max(n_points, 1), dtype=torch.int32, device=dev)
        check(lib().d3d_input_layer_export(self._h, ptr(off), ptr(idx), stream_of()))
        return off, idx[:n_points]

    def export_rules(self, kind, in_size, filter_size, stride=None, capacity=None):
        """(in, out, offset) int32 triples of a built rulebook.  kind 0 submanifold, 1 strided,
        2 deconvolution view of the strided rulebook."""
        dev = torch.device("cuda", self._dev)
        if capacity is None:
            capacity = self.getNActive(in_size) * 32 + 32
        trip = torch.empty((capacity, 3), dtype=torch.int32, device=dev)
        n = ctypes.c_long(0)
        st = ints(_size3(stride)) if stride is not None else ints([0, 0, 0])
        check(lib().d3d_export_rules(self._h, kind, ints(_size3(in_size)), ints(_size3(filter_size)), st,
                                     ptr(trip), capacity, ctypes.byref(n), stream_of()))
        return trip[:n.value]


F32, BF16 = 0, 1        # d3d_dtype


def dtype_code(t):
    """d3d_dtype of a feature tensor (fp32 or bf16 storage)."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise _lib.D3DError(f"unsupported feature dtype {t.dtype} (float32 or bfloat16)")


def stored_planes(planes, dtype):
    """Row width a tensor of `planes` channels is stored with: bf16 rows are 16 / 32 / 64 / 128 / 256 channels wide
    (zero padded), fp32 rows are exact."""
    if dtype == torch.bfloat16:
        return next(c for c in (16, 32, 64, 128, 256) if planes <= c)
    return planes


def rows_to_bf16(features, width):
    """fp32 [n, c] -> bf16 [n, width] rows, zero padded (d3d_rows_to_bf16): F.pad + .to(bfloat16) in one launch"""
    require_gpu(features)
    f = features.contiguous()
    out = torch.empty((f.shape[0], width), dtype=torch.bfloat16, device=f.device)
    check(lib().d3d_rows_to_bf16(ptr(f), f.shape[0], f.shape[1], int(width), ptr(out), stream_of()))
    return out


def n_rulebook_bits():
    return 32


def pack_weight(weight, dtype=torch.float32):
    """[fv, 1, Cin, Cout] reference layout (fp32 parameter) -> MFMA k-interleaved layout of the compute type
    (device tensor; bf16: Cin zero padded to the stored row width)."""
    require_gpu(weight)
    fv, groups, cin, cout = weight.shape
    if groups != 1:
        raise _lib.D3DError("groups != 1 is not supported")
    if dtype == torch.bfloat16:
        nbytes = lib().d3d_packed_weight_bytes(fv, cin, cout, BF16)
        if nbytes == 0:
            raise _lib.D3DError(f"unsupported bf16 conv shape fv={fv} Cin={cin} Cout={cout}")
        packed = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
        check(lib().d3d_pack_conv_weight_dt(ptr(weight.detach().float().contiguous()), fv, cin, cout, ptr(packed), BF16,
                                            stream_of()))
        return packed
    n = lib().d3d_packed_weight_floats(fv, cin, cout)
    if n == 0:
        raise _lib.D3DError(f"unsupported conv shape fv={fv} Cin={cin} Cout={cout}")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib().d3d_pack_conv_weight(ptr(weight.detach()), fv, cin, cout, ptr(packed), stream_of()))
    return packed


def _coords_to_device(coords, device):
    coords = coords.to(device=device, dtype=torch.int64, non_blocking=True)
    return coords.contiguous()


_TLS = threading.local()


def set_after_input_build(hook, prefetch_filter=None):
    """hook(metadata, spatial_size[, run_forward]) is called by this thread's next InputLayer_updateOutput calls between
    the grid build (whose site count has just been read back) and the feature pass of the input layer -- where a caller
    can start the level-0 rulebook on another stream (FPN_Net._forward_two_lane) and, through run_forward(stream), place
    the feature pass itself.  None removes it.
    prefetch_filter: the filter of the submanifold rulebook the caller will prepare first on the same stream; its
    neighbour table is then probed while the host waits for the site count (d3d_input_layer_build_prefetch)."""
    _TLS.after_input_build = hook
    _TLS.input_prefetch = None if (hook is None or prefetch_filter is None) else _size3(prefetch_filter)


def InputLayer_prepare(m):
    """d3d_input_layer_prepare: the input layer's per-site point lists, on the current stream."""
    check(lib().d3d_input_layer_prepare(m._h, stream_of()))


def InputLayer_updateOutput(m, spatial_size, input_coords, input_features, output_features,
                            batch_size, mode):
    """sparseconvnet.h:159-163.  input_coords int64 [N,3|4] on CPU or GPU."""
    require_gpu(input_features)
    dev = input_features.device
    coords = _coords_to_device(input_coords, dev)
    n, ncols = coords.shape
    na = ctypes.c_int(0)
    pre = getattr(_TLS, "input_prefetch", None)
    check(lib().d3d_input_layer_build_prefetch(m._h, ptr(coords), n, ncols, ints(_size3(spatial_size)),
                                               int(batch_size), int(mode), ints(pre) if pre is not None else None,
                                               stream_of(), ctypes.byref(na)))
    m._in_active = na.value
    planes = input_features.shape[1]
    output_features.resize_(na.value, planes)       # (allocated on the caller's stream, whichever stream fills it)
    feats = input_features.contiguous()
    done = []

    def run_forward(on=None):
        """the input layer's feature pass (d3d_input_layer_forward), on stream `on` (a torch stream) or the current one;
        the hook may call it where it suits its streams -- e.g. right behind the point lists on a side stream.
        -> the tensor being filled (None when the pass was already enqueued)"""
        if done:
            return None
        done.append(True)
        if on is None:
            check(lib().d3d_input_layer_forward(m._h, ptr(feats), planes, ptr(output_features), stream_of()))
        else:
            with torch.cuda.stream(on):
                check(lib().d3d_input_layer_forward(m._h, ptr(feats), planes, ptr(output_features), stream_of()))
        return output_features

    hook = getattr(_TLS, "after_input_build", None)
    if hook is not None:
        if hook.__code__.co_argcount >= 3:
            hook(m, spatial_size, run_forward)
        else:
            hook(m, spatial_size)
    run_forward()


def _bn_struct(bn, stats=None):
    """bn = (mean, invstd, weight, bias, leakiness) device tensors and / or stats = (fp64 [cap, 2 Cout] tensor, c_int
    that receives the rows written) -> d3d_bn_prologue (kept alive by caller)."""
    if bn is None and stats is None:
        return None
    st = _lib.BnPrologue()
    if bn is not None:
        mean, invstd, weight, bias, leak = bn
        st.mean, st.invstd = mean.data_ptr(), invstd.data_ptr()
        st.weight = weight.data_ptr() if weight is not None else None
        st.bias = bias.data_ptr() if bias is not None else None
        st.leakiness = float(leak)
    if stats is not None:
        st.out_stats, st.out_stats_cap = stats[0].data_ptr(), stats[0].shape[0]
        st.out_stats_rows = ctypes.pointer(stats[1])
    return ctypes.byref(st)


def _stats_arg(stats, n_out, cout, out):
    """stats: None, or an empty list that receives [fp64 partials tensor, c_int rows written] of this convolution"""
    if stats is None or n_out == 0:
        return None
    buf, rows = col_stats_buffer(n_out, cout, out.device)
    stats += [buf, rows]
    return buf, rows


def col_stats_buffer(n_out, cout, device):
    """-> (fp64 [cap, 2 cout] tensor, c_int) for d3d_bn_prologue.out_stats of a convolution with n_out output rows"""
    cap = ((int(n_out) + 31) // 32) * max(1, cout // 32) + 1
    return torch.empty((cap, 2 * cout), dtype=torch.float64, device=device), ctypes.c_int(0)


def stats_from_partials(partials, partial_rows, rows, eps, want_invstd=True):
    """(mean, invstd) -- or (mean, unbiased var) -- of the tensor whose producing convolution left `partials`
    (d3d_bn_stats_from_partials): what batch_mean_invstd / batch_stats compute from the tensor itself."""
    planes = partials.shape[1] // 2
    mean = torch.empty(planes, dtype=torch.float32, device=partials.device)
    other = torch.empty_like(mean)
    nbytes = _bn_scratch_bytes(planes)
    scratch = _bn_scratch(partials.device, nbytes)
    check(lib().d3d_bn_stats_from_partials(ptr(partials), int(partial_rows), int(rows), planes, float(eps),
                                           int(bool(want_invstd)), ptr(mean), ptr(other), ptr(scratch), scratch.numel(),
                                           stream_of()))
    return mean, other


_BN_SCRATCH_BYTES = {}


def _bn_scratch_bytes(planes):
    n = _BN_SCRATCH_BYTES.get(planes)
    if n is None:
        n = _BN_SCRATCH_BYTES[planes] = int(lib().d3d_bn_scratch_bytes(planes))
    return n


def batch_mean_invstd(features, eps):
    """mean(0) and powf(var_unbiased(0) + eps, -0.5) in one pass pair (eval path of batchNormalization.py:51-56
    followed by SCN/CPU/BatchNormalization.cpp:40-44)."""
    require_gpu(features)
    rows, planes = features.shape
    mean = torch.empty(planes, dtype=torch.float32, device=features.device)
    invstd = torch.empty_like(mean)
    nbytes = _bn_scratch_bytes(planes)
    scratch = _bn_scratch(features.device, nbytes)
    check(lib().d3d_bn_batch_invstd_dt(ptr(features), rows, planes, float(eps), ptr(mean), ptr(invstd), ptr(scratch),
                                       scratch.numel(), dtype_code(features), stream_of()))
    return mean, invstd


def bn_apply(features, mean, invstd, weight, bias, leakiness):
    """y = leaky(x * invstd*gamma + (beta - mean*invstd*gamma)) -- the normalisation half of
    BatchNormalization_updateOutput (SCN/CPU/BatchNormalization.cpp:46-59)."""
    require_gpu(features)
    out = torch.empty_like(features)
    check(lib().d3d_bn_apply_dt(ptr(features), ptr(out), features.shape[0], features.shape[1], ptr(mean), ptr(invstd),
                                ptr(weight), ptr(bias), float(leakiness), dtype_code(features), stream_of()))
    return out


def _conv_common(weight, packed, feats):
    """-> (filter volume, stored Cin of `feats`, Cout, packed weights of feats' dtype, d3d_dtype)"""
    fv, groups, cin, cout = weight.shape
    if packed is None or packed.dtype != feats.dtype:
        packed = pack_weight(weight, feats.dtype)
    if feats.shape[1] != stored_planes(cin, feats.dtype):
        raise _lib.D3DError(f"convolution: features have {feats.shape[1]} channels, weight expects {cin}")
    return fv, feats.shape[1], cout, packed, dtype_code(feats)


def SubmanifoldConvolution_updateOutput(spatial_size, filter_size, m, input_features,
                                        output_features, weight, bias, packed=None, residual=None, bn=None, stats=None):
    """sparseconvnet.h:99-105; returns the multiply-add count like the reference."""
    require_gpu(input_features, weight)
    if bias is not None and bias.numel():
        raise _lib.D3DError("bias is not supported (fpn_net.py builds every conv with bias=False)")
    fv, cin, cout, packed, dt = _conv_common(weight, packed, input_features)
    size, filt = _size3(spatial_size), _size3(filter_size)
    n = m.getNActive(size)
    output_features.resize_(n, cout)
    macs = ctypes.c_double(0)
    prof = PROFILER
    want = ctypes.byref(macs) if ((prof is not None and prof.learn) or COUNT_MACS) else None
    if prof is not None:   # the library records the events around the k_conv launch itself (after any rulebook build)
        t0 = prof.begin("subm", fv, cin, cout)
    check(lib().d3d_subm_conv_forward_dt(m._h, ints(size), ints(filt), ptr(input_features), cin, ptr(packed),
                                         cout, ptr(residual), ptr(output_features), dt, stream_of(), want,
                                         _bn_struct(bn, _stats_arg(stats, n, cout, output_features))))
    if prof is not None:
        prof.end(t0, "subm", fv, cin, cout, n, n, macs.value, dt)
    return macs.value


def Convolution_prepare(input_size, output_size, filter_size, filter_stride, m):
    """Builds (or finds) the strided rulebook and the output grid ahead of the convolution that uses it
    (Metadata::getRuleBook, Metadata.cpp:485-510) -> number of output sites.  One host read-back per NEW grid:
    calling this for the whole pyramid right after the input layer keeps those read-backs out of the feature pass."""
    isz, osz, filt, st = _size3(input_size), _size3(output_size), _size3(filter_size), _size3(filter_stride)
    n_out = ctypes.c_int(0)
    check(lib().d3d_conv_prepare(m._h, ints(isz), ints(osz), ints(filt), ints(st), stream_of(),
                                 ctypes.byref(n_out), None))
    return n_out.value


def SubmanifoldConvolution_prepare(spatial_size, filter_size, m):
    """Builds (or finds) the submanifold rulebook ahead of the convolutions that use it (Metadata::
    getSubmanifoldRuleBook, Metadata.cpp:430-443); no host read-back."""
    check(lib().d3d_subm_prepare(m._h, ints(_size3(spatial_size)), ints(_size3(filter_size)), stream_of(), None))


def Deconvolution_prepare(input_size, output_size, filter_size, filter_stride, m):
    """Builds (or finds) the deconvolution view (coarse `input_size` -> fine `output_size`) of the strided rulebook
    that `Convolution_prepare(output_size, input_size, ...)` made; no host read-back."""
    check(lib().d3d_deconv_prepare(m._h, ints(_size3(input_size)), ints(_size3(output_size)),
                                   ints(_size3(filter_size)), ints(_size3(filter_stride)), stream_of(), None))


def Convolution_updateOutput(input_size, output_size, filter_size, filter_stride, m, input_features,
                             output_features, weight, bias, packed=None, bn=None, stats=None):
    """sparseconvnet.h:85-91."""
    require_gpu(input_features, weight)
    if bias is not None and bias.numel():
        raise _lib.D3DError("bias is not supported")
    fv, cin, cout, packed, dt = _conv_common(weight, packed, input_features)
    isz, osz, filt, st = _size3(input_size), _size3(output_size), _size3(filter_size), _size3(filter_stride)
    n_out = ctypes.c_int(0)
    check(lib().d3d_conv_prepare(m._h, ints(isz), ints(osz), ints(filt), ints(st), stream_of(),
                                 ctypes.byref(n_out), None))
    output_features.resize_(n_out.value, cout)
    macs = ctypes.c_double(0)
    prof = PROFILER
    want = ctypes.byref(macs) if ((prof is not None and prof.learn) or COUNT_MACS) else None
    if prof is not None:
        t0 = prof.begin("conv", fv, cin, cout)
    check(lib().d3d_conv_forward_dt(m._h, ints(isz), ints(osz), ints(filt), ints(st), ptr(input_features), cin,
                                    ptr(packed), cout, ptr(output_features), dt, stream_of(), want,
                                    _bn_struct(bn, _stats_arg(stats, n_out.value, cout, output_features))))
    if prof is not None:
        prof.end(t0, "conv", fv, cin, cout, input_features.shape[0], n_out.value, macs.value, dt)
    return macs.value


def Deconvolution_updateOutput(input_size, output_size, filter_size, filter_stride, m, input_features,
                               output_features, weight, bias, packed=None, residual=None, bn=None, stats=None):
    """sparseconvnet.h:147-152: input = coarse, output = fine (rulebook of the matching Convolution)."""
    require_gpu(input_features, weight)
    if bias is not None and bias.numel():
        raise _lib.D3DError("bias is not supported")
    fv, cin, cout, packed, dt = _conv_common(weight, packed, input_features)
    isz, osz, filt, st = _size3(input_size), _size3(output_size), _size3(filter_size), _size3(filter_stride)
    n = m.getNActive(osz)
    output_features.resize_(n, cout)
    macs = ctypes.c_double(0)
    prof = PROFILER
    want = ctypes.byref(macs) if ((prof is not None and prof.learn) or COUNT_MACS) else None
    if prof is not None:
        t0 = prof.begin("deconv", fv, cin, cout)
    check(lib().d3d_deconv_forward_dt(m._h, ints(isz), ints(osz), ints(filt), ints(st), ptr(input_features),
                                      cin, ptr(packed), cout, ptr(residual), ptr(output_features), dt, stream_of(),
                                      want, _bn_struct(bn, _stats_arg(stats, n, cout, output_features))))
    if prof is not None:
        prof.end(t0, "deconv", fv, cin, cout, input_features.shape[0], n, macs.value, dt)
    return macs.value


def BatchNormalization_updateOutput(input_features, output_features, saveMean, saveInvStd, runningMean,
                                    runningVar, weight, bias, eps, momentum, train, leakiness):
    """sparseconvnet.h:21-26."""
    require_gpu(input_features, saveMean, saveInvStd, runningMean, runningVar)
    rows, planes = input_features.shape
    output_features.resize_(rows, planes)
    w = weight if (weight is not None and weight.numel()) else None
    b = bias if (bias is not None and bias.numel()) else None
    nbytes = _bn_scratch_bytes(planes)
    scratch = _bn_scratch(input_features.device, nbytes)
    check(lib().d3d_bn_forward(ptr(input_features), ptr(output_features), rows, planes, ptr(saveMean),
                               ptr(saveInvStd), ptr(runningMean), ptr(runningVar), ptr(w), ptr(b),
                               float(eps), float(momentum), int(bool(train)), float(leakiness),
                               ptr(scratch), scratch.numel(), stream_of()))


def batch_stats(features):
    """features.mean(0), features.var(0) (unbiased) as batchNormalization.py:54-55 computes them."""
    require_gpu(features)
    rows, planes = features.shape
    mean = torch.empty(planes, dtype=torch.float32, device=features.device)
    var = torch.empty_like(mean)
    nbytes = _bn_scratch_bytes(planes)
    scratch = _bn_scratch(features.device, nbytes)
    check(lib().d3d_bn_batch_stats(ptr(features), rows, planes, ptr(mean), ptr(var), ptr(scratch),
                                   scratch.numel(), stream_of()))
    return mean, var


def SparseToDense_updateOutput(spatial_size, m, input_features, output, nPlanes, batch_size=1):
    """sparseconvnet.h:214-217: dense [B, C, X, Y, Z], zero filled."""
    require_gpu(input_features)
    size = _size3(spatial_size)
    output.resize_(batch_size, nPlanes, size[0], size[1], size[2])
    check(lib().d3d_sparse_to_dense_forward(m._h, ints(size), ptr(input_features), int(nPlanes),
                                            int(batch_size), ptr(output), stream_of()))


def pack_weight_transposed(weight, flip):
    """Packing of W^T ([fv, 1, Cin, Cout] -> a conv weight with Cin'=Cout, Cout'=Cin; `flip` reverses the
    offset order, needed for the submanifold dInput)."""
    require_gpu(weight)
    fv, groups, cin, cout = weight.shape
    n = lib().d3d_packed_weight_floats(fv, cout, cin)
    if n == 0:
        raise _lib.D3DError(f"unsupported conv shape for backward fv={fv} Cin={cin} Cout={cout}")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib().d3d_pack_conv_weight_transposed(ptr(weight.detach()), fv, cin, cout, int(bool(flip)), ptr(packed),
                                                stream_of()))
    return packed


def _dinput_supported(cin):
    return cin in (32, 64, 128, 256)


def SubmanifoldConvolution_backward(spatial_size, filter_size, m, input_features, d_input_features,
                                    d_output_features, weight, d_weight, d_bias, want_d_input=True):
    """sparseconvnet.h:106-111.  d_input_features is resized and overwritten, d_weight (pre-zeroed by the
    caller, submanifoldConvolution.py backward) is accumulated into."""
    require_gpu(input_features, d_output_features, weight, d_weight)
    fv, _, cin, cout = weight.shape
    size, filt = _size3(spatial_size), _size3(filter_size)
    do = d_output_features.contiguous()
    din = None
    packed_t = None
    if want_d_input:
        if not _dinput_supported(cin):
            raise _lib.D3DError(f"dInput for Cin={cin} is not built (only the first layer has such a Cin)")
        d_input_features.resize_(input_features.shape[0], cin)
        din = d_input_features
        packed_t = pack_weight_transposed(weight, flip=True)
    check(lib().d3d_subm_conv_backward(m._h, ints(size), ints(filt), ptr(input_features), cin, ptr(packed_t), cout,
                                       ptr(do), ptr(din), ptr(d_weight), stream_of()))


def Convolution_backward(input_size, output_size, filter_size, filter_stride, m, input_features,
                         d_input_features, d_output_features, weight, d_weight, d_bias, want_d_input=True):
    """sparseconvnet.h:92-98."""
    require_gpu(input_features, d_output_features, weight, d_weight)
    fv, _, cin, cout = weight.shape
    isz, osz, filt, st = _size3(input_size), _size3(output_size), _size3(filter_size), _size3(filter_stride)
    do = d_output_features.contiguous()
    din = packed_t = None
    if want_d_input:
        d_input_features.resize_(input_features.shape[0], cin)
        din = d_input_features
        packed_t = pack_weight_transposed(weight, flip=False)
    check(lib().d3d_conv_backward(m._h, ints(isz), ints(osz), ints(filt), ints(st), ptr(input_features), cin,
                                  ptr(packed_t), cout, ptr(do), ptr(din), ptr(d_weight), stream_of()))


def Deconvolution_backward(input_size, output_size, filter_size, filter_stride, m, input_features,
                           d_input_features, d_output_features, weight, d_weight, d_bias, want_d_input=True):
    """sparseconvnet.h:153-158."""
    require_gpu(input_features, d_output_features, weight, d_weight)
    fv, _, cin, cout = weight.shape
    isz, osz, filt, st = _size3(input_size), _size3(output_size), _size3(filter_size), _size3(filter_stride)
    do = d_output_features.contiguous()
    din = packed_t = None
    if want_d_input:
        d_input_features.resize_(input_features.shape[0], cin)
        din = d_input_features
        packed_t = pack_weight_transposed(weight, flip=False)
    check(lib().d3d_deconv_backward(m._h, ints(isz), ints(osz), ints(filt), ints(st), ptr(input_features), cin,
                                    ptr(packed_t), cout, ptr(do), ptr(din), ptr(d_weight), stream_of()))


def BatchNormalization_backward(input_features, d_input_features, output_features, d_output_features,
                                saveMean, saveInvStd, runningMean, runningVar, weight, bias, d_weight, d_bias,
                                leakiness):
    """sparseconvnet.h:27-32 (d_output_features is NOT modified in place, unlike the reference)."""
    require_gpu(input_features, output_features, d_output_features, saveMean, saveInvStd)
    rows, planes = input_features.shape
    d_input_features.resize_(rows, planes)
    w = weight if (weight is not None and weight.numel()) else None
    nbytes = lib().d3d_bn_backward_scratch_bytes(planes)
    scratch = _scratch(input_features.device, nbytes)
    check(lib().d3d_bn_backward(ptr(input_features), ptr(output_features), ptr(d_output_features.contiguous()),
                                ptr(d_input_features), rows, planes, ptr(saveMean), ptr(saveInvStd), ptr(w),
                                ptr(d_weight), ptr(d_bias), float(leakiness), ptr(scratch), scratch.numel(),
                                stream_of()))


def InputLayer_updateGradInput(m, d_input_features, d_output_features):
    """sparseconvnet.h:164-167; d_input_features must be sized [n_points, planes] by the caller."""
    require_gpu(d_input_features, d_output_features)
    check(lib().d3d_input_layer_backward(m._h, ptr(d_output_features.contiguous()), d_output_features.shape[1],
                                         ptr(d_input_features), stream_of()))


def SparseToDense_updateGradInput(spatial_size, m, input_features, d_input_features, d_output):
    """sparseconvnet.h:218-222."""
    require_gpu(input_features, d_output)
    d_input_features.resize_(input_features.shape[0], input_features.shape[1])
    check(lib().d3d_sparse_to_dense_backward(m._h, ints(_size3(spatial_size)), ptr(d_output.contiguous()),
                                             input_features.shape[1], ptr(d_input_features), stream_of()))

"""nn.Module layer of the sparse backbone, mirroring the class names, constructor arguments,
parameter names/shapes and forward semantics of SparseConvNet/sparseconvnet/*.py so that
reference checkpoints load unchanged (checkpoint surface).

Every op is a torch.autograd.Function whose forward / backward call libd3d_hip.so (the reference's
sparseconvnet/*.py wrap the pybind calls the same way).
"""
import os

import torch
from torch.nn import Module, Parameter

from . import SCN


FUSE_BN_INTO_CONV = True   # inference only; results are bit-identical to the unfused path
FUSE_BN_STATS = os.environ.get("D3D_FUSE_BN_STATS", "1") != "0"   # inference only: a convolution leaves the column sums of its output per row block and the
                           # BatchNorm that follows finishes them (one small launch) instead of re-reading the tensor;
                           # same statistics up to the fp64 summation order


def toLongTensor(dimension, x):
    """sparseconvnet/utils.py:11-18."""
    if isinstance(x, torch.Tensor):
        return x.to(torch.int64)
    if isinstance(x, (list, tuple)):
        assert len(x) == dimension
        return torch.tensor([int(v) for v in x], dtype=torch.int64)
    return torch.full((dimension,), int(x), dtype=torch.int64)


def Metadata(dim, n_points=0):
    """sparseconvnet/metadata.py:15-16 (n_points sizes the HBM arena of large batches)."""
    assert dim == 3, "only Metadata_3 is built"
    return SCN.Metadata_3(SCN.arena_bytes_for(n_points))


class SparseConvNetTensor(object):
    """sparseconvnet/sparseConvNetTensor.py:12-55."""

    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features = features
        self.metadata = metadata
        self.spatial_size = spatial_size

    def get_spatial_locations(self, spatial_size=None):
        if spatial_size is None:
            spatial_size = self.spatial_size
        return self.metadata.getSpatialLocations(spatial_size)

    def to(self, device):
        self.features = self.features.to(device)
        return self

    def cuda(self):
        self.features = self.features.cuda()
        return self

    def __repr__(self):
        return (f"SparseConvNetTensor<<features.shape={tuple(self.features.shape)},"
                f"spatial size={self.spatial_size.tolist()}>>")


class _PendingBN(SparseConvNetTensor):
    """Output of an inference-mode BatchNorm whose affine + activation has not been applied yet: the next
    convolution fuses it into its gather (d3d_bn_prologue).  Any other consumer touching `.features` gets
    the materialised tensor."""

    def __init__(self, raw, bn, metadata, spatial_size):
        self._raw, self.bn, self._mat = raw, bn, None
        self.metadata, self.spatial_size = metadata, spatial_size

    @property
    def features(self):
        if self._mat is None:
            self._mat = SCN.bn_apply(self._raw, *self.bn)
        return self._mat

    @features.setter
    def features(self, v):
        self._mat = v


def _want_col_stats(feats):
    """a list for the convolution to leave the column statistics of its output in (FUSE_BN_STATS), or None"""
    return [] if (FUSE_BN_STATS and not torch.is_grad_enabled() and feats.dtype == torch.float32) else None


def _conv_output(f, stats, metadata, spatial_size):
    out = SparseConvNetTensor(f, metadata, spatial_size)
    if stats and stats[1].value > 0:
        # fp64 column sums / sums of squares per row block, the rows written, and what they describe (tensor + version)
        out.col_partials = (stats[0], stats[1].value, f, f._version)
    return out


def _col_partials(input, f):
    """the column statistics the producing convolution left for exactly this tensor, or None"""
    cp = getattr(input, "col_partials", None)
    if cp is not None and cp[2] is f and cp[3] == f._version and FUSE_BN_STATS:
        return cp
    return None


def _conv_input(input):
    """(features to read, fused-BN tuple or None)"""
    if isinstance(input, _PendingBN) and input._mat is None and input._raw.shape[1] in (32, 64, 128, 256):
        return input._raw, input.bn
    return input.features, None


def _like(inp, features, spatial_size=None):
    return SparseConvNetTensor(features, inp.metadata, inp.spatial_size if spatial_size is None else spatial_size)


class _NoGradCtx(object):
    """Stand-in for the autograd context when gradients are off."""
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        pass


def _apply(fn, *args):
    """fn.apply(*args); with gradients disabled the forward is called directly (autograd.Function.apply costs
    ~10 us of Python per call, ~50 calls per building)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoGradCtx(), *args)


class _InputLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, metadata, spatial_size, coords, batch_size, mode):
        out = feats.new_empty(0)
        SCN.InputLayer_updateOutput(metadata, spatial_size, coords, feats, out, batch_size, mode)
        ctx.meta_obj, ctx.n_points = metadata, feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, d_out):
        d_in = d_out.new_zeros((ctx.n_points, d_out.shape[1]))
        SCN.InputLayer_updateGradInput(ctx.meta_obj, d_in, d_out)
        return d_in, None, None, None, None, None


class _ConvFn(torch.autograd.Function):
    """kind 0 submanifold, 1 strided convolution, 2 deconvolution.  `residual` (optional) is added in the
    forward epilogue; its gradient is d_out."""

    @staticmethod
    def forward(ctx, feats, weight, residual, kind, metadata, in_size, out_size, filter_size, filter_stride, packed,
                bn=None, stats=None):
        out = feats.new_empty(0)
        res = None if residual is None else residual.contiguous()
        if kind == 0:
            SCN.SubmanifoldConvolution_updateOutput(in_size, filter_size, metadata, feats, out, weight, None,
                                                    packed=packed, residual=res, bn=bn, stats=stats)
        elif kind == 1:
            SCN.Convolution_updateOutput(in_size, out_size, filter_size, filter_stride, metadata, feats, out, weight,
                                         None, packed=packed, bn=bn, stats=stats)
        else:
            SCN.Deconvolution_updateOutput(in_size, out_size, filter_size, filter_stride, metadata, feats, out,
                                           weight, None, packed=packed, residual=res, bn=bn, stats=stats)
        ctx.save_for_backward(feats, weight)
        ctx.args = (kind, metadata, in_size, out_size, filter_size, filter_stride, residual is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        feats, weight = ctx.saved_tensors
        kind, metadata, in_size, out_size, filter_size, filter_stride, has_res = ctx.args
        d_out = d_out.contiguous()
        want_in = ctx.needs_input_grad[0]
        d_in = feats.new_empty(0)
        d_w = torch.zeros_like(weight)
        if kind == 0:
            SCN.SubmanifoldConvolution_backward(in_size, filter_size, metadata, feats, d_in, d_out, weight, d_w, None,
                                                want_d_input=want_in)
        elif kind == 1:
            SCN.Convolution_backward(in_size, out_size, filter_size, filter_stride, metadata, feats, d_in, d_out,
                                     weight, d_w, None, want_d_input=want_in)
        else:
            SCN.Deconvolution_backward(in_size, out_size, filter_size, filter_stride, metadata, feats, d_in, d_out,
                                       weight, d_w, None, want_d_input=want_in)
        return (d_in if want_in else None, d_w, d_out if has_res else None, None, None, None, None, None, None, None,
                None, None)


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, bias, running_mean, running_var, eps, momentum, train, leakiness):
        out = feats.new_empty(0)
        save_mean = feats.new_empty(running_mean.shape[0])
        save_invstd = feats.new_empty(running_mean.shape[0])
        SCN.BatchNormalization_updateOutput(feats, out, save_mean, save_invstd, running_mean, running_var, weight, bias,
                                            eps, momentum, train, leakiness)
        ctx.save_for_backward(feats, out, weight, save_mean, save_invstd)
        ctx.leakiness = leakiness
        return out

    @staticmethod
    def backward(ctx, d_out):
        feats, out, weight, save_mean, save_invstd = ctx.saved_tensors
        d_in = feats.new_empty(0)
        # k_bn_bwd_finish writes every channel of both (two fills per BatchNorm saved: 60 launches per training step)
        make = torch.empty_like if feats.shape[0] > 0 else torch.zeros_like
        d_w, d_b = make(save_mean), make(save_mean)
        SCN.BatchNormalization_backward(feats, d_in, out, d_out, save_mean, save_invstd, None, None, weight, None,
                                        d_w, d_b, ctx.leakiness)
        has_w = weight is not None
        return d_in, (d_w if has_w else None), (d_b if has_w else None), None, None, None, None, None, None


class _SparseToDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, metadata, spatial_size, n_planes, batch_size):
        out = feats.new_empty(0)
        SCN.SparseToDense_updateOutput(spatial_size, metadata, feats, out, n_planes, batch_size)
        ctx.save_for_backward(feats)
        ctx.args = (metadata, spatial_size)
        return out

    @staticmethod
    def backward(ctx, grad):
        (feats,) = ctx.saved_tensors
        d_in = feats.new_empty(0)
        SCN.SparseToDense_updateGradInput(ctx.args[1], ctx.args[0], feats, d_in, grad)
        return d_in, None, None, None, None


class _PackedWeightMixin(object):
    """Caches the MFMA-layout copy of `weight` until the parameter changes."""

    def _packed(self, dtype=torch.float32):
        """packed copy of the (fp32) parameter in the layout / storage type of the features it will meet"""
        w = self.weight
        tag = (w._version, w.data_ptr(), w.device)
        if getattr(self, "_packed_tag", None) != tag:
            self._packed_w = {}
            self._packed_tag = tag
        p = self._packed_w.get(dtype)
        if p is None:
            p = self._packed_w[dtype] = SCN.pack_weight(w, dtype)
        return p


class InputLayer(Module):
    """sparseconvnet/ioLayers.py:15-65.  input = [coords int64 [N,3|4], features [N,C](, batch_size)]."""

    def __init__(self, dimension, spatial_size, mode=3):
        Module.__init__(self)
        self.dimension = dimension
        self.spatial_size = toLongTensor(dimension, spatial_size)
        self.mode = mode
        self.device = None

    def to(self, device):
        self.device = device
        return self

    def forward(self, input):
        out = SparseConvNetTensor(metadata=Metadata(self.dimension, input[0].shape[0]), spatial_size=self.spatial_size)
        feats = input[1].to(self.device) if self.device else input[1]
        out.features = _apply(_InputLayerFn, feats, out.metadata, self.spatial_size, input[0],
                                           0 if len(input) == 2 else input[2], self.mode)
        return out


class SubmanifoldConvolution(Module, _PackedWeightMixin):
    """sparseconvnet/submanifoldConvolution.py:14-59; weight [fv, groups, nIn/groups, nOut/groups]."""

    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        Module.__init__(self)
        assert groups == 1 and not bias, "groups/bias are not used by FPN_Net and not built"
        self.dimension, self.groups, self.nIn, self.nOut = dimension, groups, nIn, nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = int(self.filter_size.prod().item())
        std = (2.0 * groups / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.empty(self.filter_volume, groups, nIn // groups, nOut // groups).normal_(0, std))

    def forward(self, input, residual=None):
        feats, bn = _conv_input(input)
        assert feats.nelement() == 0 or feats.size(1) == SCN.stored_planes(self.nIn, feats.dtype), (self.nIn, self.nOut)
        stats = _want_col_stats(feats)
        f = _apply(_ConvFn, feats, self.weight, None if residual is None else residual.features, 0,
                          input.metadata, input.spatial_size, input.spatial_size, self.filter_size, None,
                          self._packed(feats.dtype), bn, stats)
        return _conv_output(f, stats, input.metadata, input.spatial_size)

    def input_spatial_size(self, out_size):
        return out_size


class Convolution(Module, _PackedWeightMixin):
    """sparseconvnet/convolution.py:13-70."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        Module.__init__(self)
        assert groups == 1 and not bias
        self.dimension, self.groups, self.nIn, self.nOut = dimension, groups, nIn, nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = int(self.filter_size.prod().item())
        self.filter_stride = toLongTensor(dimension, filter_stride)
        std = (2.0 * groups / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.empty(self.filter_volume, groups, nIn // groups, nOut // groups).normal_(0, std))

    def forward(self, input):
        feats, bn = _conv_input(input)
        assert feats.nelement() == 0 or feats.size(1) == SCN.stored_planes(self.nIn, feats.dtype)
        out_size = self._out_size(input.spatial_size)
        stats = _want_col_stats(feats)
        f = _apply(_ConvFn, feats, self.weight, None, 1, input.metadata, input.spatial_size, out_size,
                          self.filter_size, self.filter_stride, self._packed(feats.dtype), bn, stats)
        return _conv_output(f, stats, input.metadata, out_size)

    def _out_size(self, in_size):
        """(in - filter) // stride + 1, checked to tile the input exactly; remembered per input size (a pass asks for
        the same few sizes every time, and each LongTensor operation costs the launch thread 2-3 us)"""
        key = SCN._size3(in_size)
        cache = self.__dict__.setdefault("_out_sizes", {})
        out = cache.get(key)
        if out is None:
            out = (in_size - self.filter_size) // self.filter_stride + 1
            assert ((out - 1) * self.filter_stride + self.filter_size == in_size).all(), \
                (in_size, out, self.filter_size, self.filter_stride)
            cache[key] = out
        return out

    def input_spatial_size(self, out_size):
        return (out_size - 1) * self.filter_stride + self.filter_size


class Deconvolution(Module, _PackedWeightMixin):
    """sparseconvnet/deconvolution.py:13-47."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        Module.__init__(self)
        assert groups == 1 and not bias
        self.dimension, self.groups, self.nIn, self.nOut = dimension, groups, nIn, nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = int(self.filter_size.prod().item())
        self.filter_stride = toLongTensor(dimension, filter_stride)
        std = (2.0 * groups / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.empty(self.filter_volume, groups, nIn // groups, nOut // groups).normal_(0, std))

    def forward(self, input, residual=None):
        feats, bn = _conv_input(input)
        assert feats.nelement() == 0 or feats.size(1) == SCN.stored_planes(self.nIn, feats.dtype)
        key = SCN._size3(input.spatial_size)
        cache = self.__dict__.setdefault("_out_sizes", {})
        out_size = cache.get(key)
        if out_size is None:         # remembered per input size: three LongTensor operations per call otherwise
            out_size = cache[key] = (input.spatial_size - 1) * self.filter_stride + self.filter_size
        stats = _want_col_stats(feats)
        f = _apply(_ConvFn, feats, self.weight, None if residual is None else residual.features, 2,
                          input.metadata, input.spatial_size, out_size, self.filter_size, self.filter_stride,
                          self._packed(feats.dtype), bn, stats)
        return _conv_output(f, stats, input.metadata, out_size)

    def input_spatial_size(self, out_size):
        return (out_size - self.filter_size) // self.filter_stride + 1


class BatchNormalization(Module):
    """sparseconvnet/batchNormalization.py:13-68.  `momentum` is the RETENTION factor of the
    running statistics (SCN/CPU/BatchNormalization.cpp:32-36)."""

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1, track_running_stats=True):
        Module.__init__(self)
        self.nPlanes, self.eps, self.momentum, self.affine, self.leakiness = nPlanes, eps, momentum, affine, leakiness
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        if affine:
            self.weight = Parameter(torch.ones(nPlanes))
            self.bias = Parameter(torch.zeros(nPlanes))
        self.track_running_stats = track_running_stats

    def forward(self, input):
        f = input.features
        assert f.nelement() == 0 or f.size(1) == self.nPlanes, (self.nPlanes, f.shape)
        if not (self.training or self.track_running_stats) and not torch.is_grad_enabled() and FUSE_BN_INTO_CONV:
            # inference with batch statistics (batchNormalization.py:53-55): hand (mean, invstd, gamma, beta, leak)
            # to the consuming convolution instead of writing the normalised tensor
            cp = _col_partials(input, f)
            if cp is not None:
                mean, invstd = SCN.stats_from_partials(cp[0], cp[1], f.shape[0], self.eps, want_invstd=True)
            else:
                mean, invstd = SCN.batch_mean_invstd(f, self.eps)
            return _PendingBN(f, (mean, invstd, self.weight if self.affine else None,
                                  self.bias if self.affine else None, self.leakiness), input.metadata,
                              input.spatial_size)
        if f.dtype != torch.float32:    # bf16 storage: inference with batch statistics only (the deferred form below)
            mean, invstd = SCN.batch_mean_invstd(f, self.eps)
            return _like(input, SCN.bn_apply(f, mean, invstd, self.weight if self.affine else None,
                                             self.bias if self.affine else None, self.leakiness))
        if self.training or self.track_running_stats:
            mean, var = self.running_mean, self.running_var
        else:  # batchNormalization.py:53-55: batch statistics stand in for the running ones
            cp = _col_partials(input, f) if not torch.is_grad_enabled() else None
            if cp is not None:
                mean, var = SCN.stats_from_partials(cp[0], cp[1], f.shape[0], self.eps, want_invstd=False)
            else:
                mean, var = SCN.batch_stats(f.detach())
        y = _apply(_BatchNormFn, f, self.weight if self.affine else None, self.bias if self.affine else None,
                               mean, var, self.eps, self.momentum, self.training, self.leakiness)
        return _like(input, y)

    def input_spatial_size(self, out_size):
        return out_size


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0, track_running_stats)


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, leakiness, track_running_stats)


class Sequential(torch.nn.Sequential):
    """sparseconvnet/sequential.py:9-27."""

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        for m in reversed(self._modules):
            out_size = self._modules[m].input_spatial_size(out_size)
        return out_size


class Identity(Module):
    def forward(self, input):
        return input

    def input_spatial_size(self, out_size):
        return out_size


class ConcatTable(torch.nn.Sequential):
    """sparseconvnet/tables.py:44-55: applies every child to the same input."""

    def forward(self, input):
        return [module(input) for module in self._modules.values()]

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        return self._modules["0"].input_spatial_size(out_size)


def add_feature_planes(inputs):
    """sparseconvnet/utils.py:61-66 (two operands on this path)."""
    from .._lib import check, lib, ptr, stream_of
    assert len(inputs) >= 2
    acc = inputs[0].features
    if torch.is_grad_enabled() and any(i.features.requires_grad for i in inputs):
        for other in inputs[1:]:
            acc = acc + other.features
        return _like(inputs[0], acc)
    for other in inputs[1:]:
        if acc.dtype != torch.float32:          # bf16 storage: fp32 sum, one rounding
            acc = (acc.float() + other.features.float()).to(acc.dtype)
            continue
        out = torch.empty_like(acc)
        check(lib().d3d_add(ptr(acc), ptr(other.features.contiguous()), ptr(out), acc.numel(), stream_of()))
        acc = out
    return _like(inputs[0], acc)


class AddTable(torch.nn.Sequential):
    """sparseconvnet/tables.py:28-41."""

    def forward(self, input):
        return add_feature_planes(input)

    def input_spatial_size(self, out_size):
        return out_size


class OutputLayer(Module):
    """sparseconvnet/ioLayers.py (OutputLayer): only constructed, never called, by FPN_Net."""

    def __init__(self, dimension):
        Module.__init__(self)
        self.dimension = dimension

    def forward(self, input):
        raise NotImplementedError("OutputLayer is unused on the detection path (fpn_net.py:47-49)")


class SparseToDense(Module):
    """sparseconvnet/sparseToDense.py: dense [B, C, X, Y, Z]."""

    def __init__(self, dimension, nPlanes):
        Module.__init__(self)
        self.dimension, self.nPlanes = dimension, nPlanes

    def forward(self, input, batch_size=None):
        if batch_size is None:
            loc = input.get_spatial_locations()
            batch_size = int(loc[:, 3].max().item()) + 1 if loc.shape[0] else 1
        return _apply(_SparseToDenseFn, input.features, input.metadata, input.spatial_size, self.nPlanes, batch_size)

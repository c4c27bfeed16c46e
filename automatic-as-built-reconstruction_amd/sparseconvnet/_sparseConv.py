"""Shared machinery of the three sparse convolution layers (extension of this package's own making: the reference writes
each layer out in full -- submanifoldConvolution.py:14-113, convolution.py:14-90, deconvolution.py:13-87 -- with its own
autograd Function).  One Module base + ONE autograd Function, parameterised by a `_Kind` record that says how a layer maps
spatial sizes and which pair of `SCN` entry points serves it.  What stays the reference's, because checkpoints, model code
and `print(model)` depend on it: class names, constructor arguments, the attributes `dimension / groups / nIn / nOut /
filter_size / filter_volume / filter_stride`, parameters `weight` [vol, groups, nIn/g, nOut/g] and `bias`, He-style
initialisation N(0, 2 g / (nIn vol)), `input_spatial_size`, and the text of `__repr__`."""
import collections

import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

import sparseconvnet
from . import SCN
from .sparseConvNetTensor import SparseConvNetTensor
from .utils import toLongTensor, optionalTensor, optionalTensorReturn

# name in __repr__ | has a stride | output size from (input size, filter, stride) | its inverse | SCN forward | SCN backward
_Kind = collections.namedtuple("_Kind", "label strided out_size in_size fwd bwd")


def _same(sz, f, s):
    return sz


KINDS = {
    "subm": _Kind("SubmanifoldConvolution", False, _same, _same,
                  lambda g, md, x, y, w, b, pk: SCN.SubmanifoldConvolution_updateOutput(g[0], g[2], md, x, y, w, b, pack_t=pk),
                  lambda g, md, x, dx, dy, w, dw, db, pk, need: SCN.SubmanifoldConvolution_backward(
                      g[0], g[2], md, x, dx, dy, w, dw, db, pack_t=pk, need_d_input=need)),
    # floor division: the reference's `/` on LongTensors (convolution.py:35-36) under the PyTorch 1.x it was written for
    "conv": _Kind("Convolution", True, lambda sz, f, s: (sz - f) // s + 1, lambda sz, f, s: (sz - 1) * s + f,
                  lambda g, md, x, y, w, b, pk: SCN.Convolution_updateOutput(g[0], g[1], g[2], g[3], md, x, y, w, b, pack_t=pk),
                  lambda g, md, x, dx, dy, w, dw, db, pk, need: SCN.Convolution_backward(
                      g[0], g[1], g[2], g[3], md, x, dx, dy, w, dw, db, pack_t=pk, need_d_input=need)),
    # the "convolution reversing" transpose: restores the finer grid the Metadata still holds
    "deconv": _Kind("Deconvolution", True, lambda sz, f, s: (sz - 1) * s + f, lambda sz, f, s: (sz - f) // s + 1,
                    lambda g, md, x, y, w, b, pk: SCN.Deconvolution_updateOutput(g[0], g[1], g[2], g[3], md, x, y, w, b, pack_t=pk),
                    lambda g, md, x, dx, dy, w, dw, db, pk, need: SCN.Deconvolution_backward(
                        g[0], g[1], g[2], g[3], md, x, dx, dy, w, dw, db, pack_t=pk, need_d_input=need)),
}


def _dims(t):
    """'3' for an isotropic LongTensor, '(1,1,8)' otherwise"""
    v = [int(i) for i in t]
    return str(v[0]) if min(v) == max(v) else "(" + ",".join(map(str, v)) + ")"


class SparseConvModule(Module):
    kind = None        # key of KINDS, set by the three public classes

    def _setup(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups):
        Module.__init__(self)
        self.dimension, self.groups, self.nIn, self.nOut = dimension, groups, nIn, nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = self.filter_size.prod().item()
        if KINDS[self.kind].strided:
            self.filter_stride = toLongTensor(dimension, filter_stride)
        std = (2.0 * groups / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.Tensor(self.filter_volume, groups, nIn // groups, nOut // groups).normal_(0, std))
        if bias:
            self.bias = Parameter(torch.Tensor(nOut).zero_())

    def _stride(self):
        return self.filter_stride if KINDS[self.kind].strided else None

    def forward(self, input):
        k = KINDS[self.kind]
        assert input.features.nelement() == 0 or input.features.size(1) == self.nIn, (self.nIn, self.nOut, input)
        out_sz = k.out_size(input.spatial_size, self.filter_size, self._stride())
        if self.kind == "conv":
            assert (k.in_size(out_sz, self.filter_size, self.filter_stride) == input.spatial_size).all(), (
                input.spatial_size, out_sz, self.filter_size, self.filter_stride)
        feats = SparseConvFunction.apply(input.features, self.weight, optionalTensor(self, "bias"), input.metadata,
                                         self.kind, (input.spatial_size, out_sz, self.filter_size, self._stride()))
        return SparseConvNetTensor(feats, input.metadata, out_sz)

    def input_spatial_size(self, out_size):
        return KINDS[self.kind].in_size(out_size, self.filter_size, self._stride())

    def __repr__(self):
        k = KINDS[self.kind]
        if not k.strided:
            return "%s %d->%d C%s" % (k.label, self.nIn, self.nOut, _dims(self.filter_size))
        iso = "(" not in _dims(self.filter_size) and "(" not in _dims(self.filter_stride)
        f, s = _dims(self.filter_size), _dims(self.filter_stride)
        if not iso:      # the reference brackets both as soon as one of them is anisotropic
            f = "(" + ",".join(str(int(i)) for i in self.filter_size) + ")"
            s = "(" + ",".join(str(int(i)) for i in self.filter_stride) + ")"
        return "%s %d->%d C%s/%s" % (k.label, self.nIn, self.nOut, f, s)


class SparseConvFunction(Function):
    """forward = the layer's SCN *_updateOutput, backward = its *_backward (input gradient only when somebody needs it;
    the weight-gradient kernels write every element of dW, so nothing is pre-zeroed -- the reference zeroes because its
    CUDA path accumulates with atomicAdd, SCN/CUDA/Convolution.cu:318)."""

    @staticmethod
    def forward(ctx, x, weight, bias, metadata, kind, geom):
        ctx.scn_md, ctx.kind, ctx.geom = metadata, kind, geom
        # the input-gradient layout of the weights is packed in the forward pack's launch when a backward pass through
        # this layer will want it
        ctx.pack_t = [] if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x, weight, bias)
        y = x.new()
        sparseconvnet.forward_pass_multiplyAdd_count += KINDS[kind].fwd(geom, metadata, x, y, weight, bias, ctx.pack_t)
        sparseconvnet.forward_pass_hidden_states += y.nelement()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        need = ctx.needs_input_grad[0]
        dx, dw, db = dy.new(), torch.empty_like(weight), torch.zeros_like(bias)
        KINDS[ctx.kind].bwd(ctx.geom, ctx.scn_md, x, dx, dy.contiguous(), weight, dw, db, ctx.pack_t, need)
        return (dx if need else None), dw, optionalTensorReturn(db), None, None, None

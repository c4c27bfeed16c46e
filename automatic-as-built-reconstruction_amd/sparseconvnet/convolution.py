"""Convolution (reference: SparseConvNet/sparseconvnet/convolution.py).  The output spatial size
uses floor division: the reference's `/` on LongTensors (:35-36) is integer division under the
PyTorch 1.x it was written for."""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

import sparseconvnet
from . import SCN
from .utils import toLongTensor, optionalTensor, optionalTensorReturn
from .sparseConvNetTensor import SparseConvNetTensor


class Convolution(Module):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        Module.__init__(self)
        self.dimension = dimension
        self.groups = groups
        self.nIn = nIn
        self.nOut = nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = self.filter_size.prod().item()
        self.filter_stride = toLongTensor(dimension, filter_stride)
        std = (2.0 * groups / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.Tensor(self.filter_volume, groups, nIn // groups, nOut // groups).normal_(0, std))
        if bias:
            self.bias = Parameter(torch.Tensor(nOut).zero_())

    def forward(self, input):
        assert input.features.nelement() == 0 or input.features.size(1) == self.nIn
        output = SparseConvNetTensor()
        output.metadata = input.metadata
        output.spatial_size = (input.spatial_size - self.filter_size) // self.filter_stride + 1
        assert ((output.spatial_size - 1) * self.filter_stride + self.filter_size == input.spatial_size).all(), (
            input.spatial_size, output.spatial_size, self.filter_size, self.filter_stride)
        output.features = ConvolutionFunction.apply(
            input.features, self.weight, optionalTensor(self, "bias"), input.metadata, input.spatial_size,
            output.spatial_size, self.dimension, self.filter_size, self.filter_stride)
        return output

    def __repr__(self):
        s = "Convolution " + str(self.nIn) + "->" + str(self.nOut) + " C"
        if self.filter_size.max().item() == self.filter_size.min().item() and \
                self.filter_stride.max().item() == self.filter_stride.min().item():
            s = s + str(self.filter_size[0].item()) + "/" + str(self.filter_stride[0].item())
        else:
            s = s + "(" + ",".join(str(i.item()) for i in self.filter_size) + ")/(" + \
                ",".join(str(i.item()) for i in self.filter_stride) + ")"
        return s

    def input_spatial_size(self, out_size):
        return (out_size - 1) * self.filter_stride + self.filter_size


class ConvolutionFunction(Function):
    @staticmethod
    def forward(ctx, input_features, weight, bias, input_metadata, input_spatial_size, output_spatial_size,
                dimension, filter_size, filter_stride):
        output_features = input_features.new()
        # the input-gradient layout of the weights is packed together with the forward one (one launch)
        # when a backward pass through this layer will need it
        ctx.pack_t = [] if ctx.needs_input_grad[0] else None
        ctx.input_metadata = input_metadata
        ctx.dimension = dimension
        ctx.geom = (input_spatial_size, output_spatial_size, filter_size, filter_stride)
        ctx.save_for_backward(input_features, weight, bias)
        sparseconvnet.forward_pass_multiplyAdd_count += SCN.Convolution_updateOutput(
            input_spatial_size, output_spatial_size, filter_size, filter_stride, input_metadata, input_features,
            output_features, weight, bias, pack_t=ctx.pack_t)
        sparseconvnet.forward_pass_hidden_states += output_features.nelement()
        return output_features

    @staticmethod
    def backward(ctx, grad_output):
        input_features, weight, bias = ctx.saved_tensors
        input_spatial_size, output_spatial_size, filter_size, filter_stride = ctx.geom
        grad_input = grad_output.new()
        # the weight-gradient kernel writes every element (the reference pre-zeroes because its
        # CUDA path accumulates with atomicAdd, Convolution.cu:318); no fill launch needed
        grad_weight = torch.empty_like(weight)
        grad_bias = torch.zeros_like(bias)
        SCN.Convolution_backward(input_spatial_size, output_spatial_size, filter_size, filter_stride,
                                 ctx.input_metadata, input_features, grad_input, grad_output.contiguous(), weight,
                                 grad_weight, grad_bias, pack_t=ctx.pack_t, need_d_input=ctx.needs_input_grad[0])
        return (grad_input if ctx.needs_input_grad[0] else None), grad_weight, optionalTensorReturn(grad_bias), None, None, None, None, None, None

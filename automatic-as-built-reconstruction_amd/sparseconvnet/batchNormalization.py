"""BatchNormalization / BatchNormReLU / BatchNormLeakyReLU
(reference: SparseConvNet/sparseconvnet/batchNormalization.py:13-172).  Same constructor
arguments, parameter / buffer names (weight, bias, running_mean, running_var) and the same
`track_running_stats=False` evaluation rule (:51-56)."""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from . import SCN
from .utils import optionalTensor, optionalTensorReturn
from .sparseConvNetTensor import SparseConvNetTensor


class BatchNormalization(Module):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1, track_running_stats=True):
        Module.__init__(self)
        self.nPlanes = nPlanes
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        self.leakiness = leakiness
        self.register_buffer("running_mean", torch.Tensor(nPlanes).fill_(0))
        self.register_buffer("running_var", torch.Tensor(nPlanes).fill_(1))
        if affine:
            self.weight = Parameter(torch.Tensor(nPlanes).fill_(1))
            self.bias = Parameter(torch.Tensor(nPlanes).fill_(0))
        self.track_running_stats = track_running_stats

    def forward(self, input):
        assert input.features.nelement() == 0 or input.features.size(1) == self.nPlanes, (
            self.nPlanes, input.features.shape)
        output = SparseConvNetTensor()
        output.metadata = input.metadata
        output.spatial_size = input.spatial_size
        if self.training or self.track_running_stats:
            _mean = self.running_mean
            _var = self.running_var
        else:
            # statistics buffers are fp32 whatever the feature storage (the kernels read `planes` floats)
            _mean = input.features.float().mean(0)
            _var = input.features.float().var(0)
        output.features = BatchNormalizationFunction.apply(
            input.features, optionalTensor(self, "weight"), optionalTensor(self, "bias"), _mean, _var, self.eps,
            self.momentum, self.training, self.leakiness)
        return output

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        s = "BatchNorm(" + str(self.nPlanes) + ",eps=" + str(self.eps) + ",momentum=" + str(self.momentum) + \
            ",affine=" + str(self.affine)
        if self.leakiness > 0:
            s = s + ",leakiness=" + str(self.leakiness)
        return s + ")"


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0, track_running_stats)

    def __repr__(self):
        return "BatchNormReLU(" + str(self.nPlanes) + ",eps=" + str(self.eps) + ",momentum=" + \
            str(self.momentum) + ",affine=" + str(self.affine) + ")"


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, leakiness, track_running_stats)

    def __repr__(self):
        return "BatchNormLeakyReLU(" + str(self.nPlanes) + ",eps=" + str(self.eps) + ",momentum=" + \
            str(self.momentum) + ",affine=" + str(self.affine) + ",leakiness=" + str(self.leakiness) + ")"


class BatchNormalizationFunction(Function):
    @staticmethod
    def forward(ctx, input_features, weight, bias, running_mean, running_var, eps, momentum, train, leakiness):
        ctx.nPlanes = running_mean.shape[0]
        ctx.train = train
        ctx.leakiness = leakiness
        output_features = input_features.new()
        # statistics are fp32 whatever the feature storage type is
        saveMean = torch.empty(ctx.nPlanes, dtype=torch.float32, device=input_features.device)
        saveInvStd = torch.empty(ctx.nPlanes, dtype=torch.float32, device=input_features.device)
        SCN.BatchNormalization_updateOutput(input_features, output_features, saveMean, saveInvStd, running_mean,
                                            running_var, weight, bias, eps, momentum, ctx.train, ctx.leakiness)
        ctx.save_for_backward(input_features, output_features, weight, bias, running_mean, running_var, saveMean,
                              saveInvStd)
        return output_features

    @staticmethod
    def backward(ctx, grad_output):
        input_features, output_features, weight, bias, running_mean, running_var, saveMean, saveInvStd = \
            ctx.saved_tensors
        assert ctx.train
        grad_input = grad_output.new()
        grad_weight = torch.empty_like(weight)  # fully written by the backward finalize kernel
        grad_bias = torch.empty_like(bias)
        SCN.BatchNormalization_backward(input_features, grad_input, output_features, grad_output.contiguous(),
                                        saveMean, saveInvStd, running_mean, running_var, weight, bias, grad_weight,
                                        grad_bias, ctx.leakiness)
        return grad_input, optionalTensorReturn(grad_weight), optionalTensorReturn(grad_bias), None, None, None, \
            None, None, None

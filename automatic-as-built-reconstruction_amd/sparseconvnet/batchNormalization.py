"""BatchNormalization / BatchNormReLU / BatchNormLeakyReLU over the rows of a sparse feature matrix, the activation fused
(reference: SparseConvNet/sparseconvnet/batchNormalization.py:13-172).  Kept from the reference because checkpoints and
model code depend on it: class names, constructor arguments, `nPlanes / eps / momentum / affine / leakiness /
track_running_stats`, parameters `weight`, `bias`, buffers `running_mean`, `running_var`, the rule that a layer with
`track_running_stats=False` normalises by the batch's own statistics in evaluation mode (:51-56), and the `__repr__`
text.  leakiness: 1 = no activation, 0 = ReLU, in between = leaky ReLU."""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from . import SCN
from .sparseConvNetTensor import SparseConvNetTensor
from .utils import optionalTensor, optionalTensorReturn


class BatchNormalization(Module):
    _label, _show_leak = "BatchNorm", None       # None: print the leakiness only when there is an activation

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1, track_running_stats=True):
        Module.__init__(self)
        self.nPlanes, self.eps, self.momentum, self.affine, self.leakiness = nPlanes, eps, momentum, affine, leakiness
        self.register_buffer("running_mean", torch.Tensor(nPlanes).fill_(0))
        self.register_buffer("running_var", torch.Tensor(nPlanes).fill_(1))
        if affine:
            self.weight = Parameter(torch.Tensor(nPlanes).fill_(1))
            self.bias = Parameter(torch.Tensor(nPlanes).fill_(0))
        self.track_running_stats = track_running_stats

    def _statistics(self, features):
        """the (mean, variance) pair the kernel normalises by in evaluation mode / updates in training mode; always fp32,
        whatever the feature storage"""
        if self.training or self.track_running_stats:
            return self.running_mean, self.running_var
        f = features.float()
        return f.mean(0), f.var(0)

    def forward(self, input):
        x = input.features
        assert x.nelement() == 0 or x.size(1) == self.nPlanes, (self.nPlanes, x.shape)
        mean, var = self._statistics(x)
        y = BatchNormFunction.apply(x, optionalTensor(self, "weight"), optionalTensor(self, "bias"), mean, var,
                                    (self.eps, self.momentum, self.training, self.leakiness))
        return SparseConvNetTensor(y, input.metadata, input.spatial_size)

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        show = self.leakiness > 0 if self._show_leak is None else self._show_leak
        return "%s(%s,eps=%s,momentum=%s,affine=%s%s)" % (self._label, self.nPlanes, self.eps, self.momentum, self.affine,
                                                         ",leakiness=" + str(self.leakiness) if show else "")


class BatchNormReLU(BatchNormalization):
    _label, _show_leak = "BatchNormReLU", False

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0, track_running_stats)


class BatchNormLeakyReLU(BatchNormalization):
    _label, _show_leak = "BatchNormLeakyReLU", True

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333, track_running_stats=True):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, leakiness, track_running_stats)


class BatchNormFunction(Function):
    """SCN.BatchNormalization_updateOutput / _backward; the batch mean and 1/std of a training pass are kept (fp32) for
    the backward pass; the backward finalize kernel writes every element of the two parameter gradients"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, cfg):
        eps, momentum, train, leakiness = cfg
        ctx.train, ctx.leakiness = train, leakiness
        n = running_mean.shape[0]
        y = x.new()
        mean = torch.empty(n, dtype=torch.float32, device=x.device)
        inv_std = torch.empty(n, dtype=torch.float32, device=x.device)
        SCN.BatchNormalization_updateOutput(x, y, mean, inv_std, running_mean, running_var, weight, bias, eps, momentum,
                                            train, leakiness)
        ctx.save_for_backward(x, y, weight, bias, running_mean, running_var, mean, inv_std)
        return y

    @staticmethod
    def backward(ctx, dy):
        assert ctx.train
        x, y, weight, bias, running_mean, running_var, mean, inv_std = ctx.saved_tensors
        dx, dw, db = dy.new(), torch.empty_like(weight), torch.empty_like(bias)
        SCN.BatchNormalization_backward(x, dx, y, dy.contiguous(), mean, inv_std, running_mean, running_var, weight, bias,
                                        dw, db, ctx.leakiness)
        return dx, optionalTensorReturn(dw), optionalTensorReturn(db), None, None, None

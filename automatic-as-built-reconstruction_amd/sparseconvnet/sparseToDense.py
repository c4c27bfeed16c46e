"""SparseToDense: a sparse hidden layer written out as a dense [batch, planes, X, Y, Z] tensor, zeros where no site is
active (reference: SparseConvNet/sparseconvnet/sparseToDense.py:14-73; same class name, arguments and output layout)."""
from torch.autograd import Function
from torch.nn import Module

from . import SCN


class SparseToDense(Module):
    def __init__(self, dimension, nPlanes):
        Module.__init__(self)
        self.dimension, self.nPlanes = dimension, nPlanes

    def forward(self, input):
        return _Densify.apply(input.features, input.metadata, input.spatial_size, self.nPlanes)

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        return "SparseToDense(%s,%s)" % (self.dimension, self.nPlanes)


class _Densify(Function):
    @staticmethod
    def forward(ctx, x, metadata, spatial_size, nPlanes):
        ctx.where = (metadata, spatial_size)
        ctx.save_for_backward(x)
        dense = x.new()
        SCN.SparseToDense_updateOutput(spatial_size, metadata, x, dense, nPlanes)
        return dense

    @staticmethod
    def backward(ctx, d_dense):
        x, = ctx.saved_tensors
        metadata, spatial_size = ctx.where
        dx = d_dense.new()
        SCN.SparseToDense_updateGradInput(spatial_size, metadata, x, dx, d_dense.contiguous())
        return dx, None, None, None


SparseToDenseFunction = _Densify     # the reference's name for the autograd node

"""SparseToDense (reference: SparseConvNet/sparseconvnet/sparseToDense.py): sparse hidden layer ->
dense [batch, planes, X, Y, Z] tensor."""
from torch.autograd import Function
from torch.nn import Module

from . import SCN


class SparseToDenseFunction(Function):
    @staticmethod
    def forward(ctx, input_features, input_metadata, spatial_size, dimension, nPlanes):
        ctx.input_metadata = input_metadata
        ctx.dimension = dimension
        ctx.spatial_size = spatial_size
        ctx.save_for_backward(input_features)
        output = input_features.new()
        SCN.SparseToDense_updateOutput(spatial_size, input_metadata, input_features, output, nPlanes)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_input = grad_output.new()
        input_features, = ctx.saved_tensors
        SCN.SparseToDense_updateGradInput(ctx.spatial_size, ctx.input_metadata, input_features, grad_input,
                                          grad_output.contiguous())
        return grad_input, None, None, None, None


class SparseToDense(Module):
    def __init__(self, dimension, nPlanes):
        Module.__init__(self)
        self.dimension = dimension
        self.nPlanes = nPlanes

    def forward(self, input):
        return SparseToDenseFunction.apply(input.features, input.metadata, input.spatial_size, self.dimension,
                                           self.nPlanes)

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        return "SparseToDense(" + str(self.dimension) + "," + str(self.nPlanes) + ")"

"""InputLayer (reference: SparseConvNet/sparseconvnet/ioLayers.py:15-65,163-195).

Same constructor, same (coords, features[, batch_size]) input tuple, same modes.  The
reference forces `coords` to the host (`input[0].cpu().long()`, :60) because its hash grid is
a host structure; here the grid is built on the device, so device-resident coordinates are
used as they are and host coordinates are uploaded once."""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import SCN
from .utils import toLongTensor
from .sparseConvNetTensor import SparseConvNetTensor
from .metadata import Metadata


class InputLayer(Module):
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
        output = SparseConvNetTensor(metadata=Metadata(self.dimension), spatial_size=self.spatial_size)
        output.features = InputLayerFunction.apply(
            self.dimension, output.metadata, self.spatial_size, input[0].long(),
            input[1].to(self.device) if self.device else input[1], 0 if len(input) == 2 else input[2],
            self.mode)
        return output


class InputLayerFunction(Function):
    @staticmethod
    def forward(ctx, dimension, metadata, spatial_size, coords, input_features, batch_size, mode):
        output_features = input_features.new()
        ctx.dimension = dimension
        ctx.metadata_ = metadata
        SCN.InputLayer_updateOutput(metadata, spatial_size, coords, input_features.contiguous(), output_features,
                                    batch_size, mode)
        return output_features

    @staticmethod
    def backward(ctx, grad_output):
        grad_input = grad_output.new()
        SCN.InputLayer_updateGradInput(ctx.metadata_, grad_input, grad_output.contiguous())
        return None, None, None, None, grad_input, None, None

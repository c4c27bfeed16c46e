"""NetworkInNetwork: 1x1 convolution == a dense [V,nIn] x [nIn,nOut] product (reference:
SparseConvNet/sparseconvnet/networkInNetwork.py; its GPU path is a plain library GEMM too,
SCN/CUDA/NetworkInNetwork.cpp:16-48).  Only instantiated by FPN_Net when a residual block
changes width, which the default configuration never does (fpn_net.py:62)."""
import torch
from torch.nn import Module, Parameter

from .sparseConvNetTensor import SparseConvNetTensor


class NetworkInNetwork(Module):
    def __init__(self, nIn, nOut, bias):
        Module.__init__(self)
        self.nIn = nIn
        self.nOut = nOut
        std = (2.0 / nIn) ** 0.5
        self.weight = Parameter(torch.Tensor(nIn, nOut).normal_(0, std))
        if bias:
            self.bias = Parameter(torch.Tensor(nOut).zero_())

    def forward(self, input):
        assert input.features.nelement() == 0 or input.features.size(1) == self.nIn
        output = SparseConvNetTensor()
        output.metadata = input.metadata
        output.spatial_size = input.spatial_size
        f = input.features @ self.weight
        if hasattr(self, "bias"):
            f = f + self.bias
        output.features = f
        return output

    def input_spatial_size(self, out_size):
        return out_size

"""JoinTable / AddTable / ConcatTable (reference: SparseConvNet/sparseconvnet/tables.py:13-55)."""
import torch

from .sparseConvNetTensor import SparseConvNetTensor
from .utils import _sum_features


class JoinTable(torch.nn.Sequential):
    def forward(self, input):
        output = SparseConvNetTensor()
        output.metadata = input[0].metadata
        output.spatial_size = input[0].spatial_size
        output.features = torch.cat([i.features for i in input], 1) if input[0].features.numel() else \
            input[0].features
        return output

    def input_spatial_size(self, out_size):
        return out_size


class AddTable(torch.nn.Sequential):
    def forward(self, input):
        output = SparseConvNetTensor()
        output.metadata = input[0].metadata
        output.spatial_size = input[0].spatial_size
        output.features = _sum_features(input)
        return output

    def input_spatial_size(self, out_size):
        return out_size


class ConcatTable(torch.nn.Sequential):
    def forward(self, input):
        return [module(input) for module in self._modules.values()]

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        return self._modules["0"].input_spatial_size(out_size)

"""Sequential container (reference: SparseConvNet/sparseconvnet/sequential.py:9-16)."""
import torch


class Sequential(torch.nn.Sequential):
    def input_spatial_size(self, out_size):
        for m in reversed(self._modules):
            out_size = self._modules[m].input_spatial_size(out_size)
        return out_size

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def insert(self, index, module):
        for i in range(len(self._modules), index, -1):
            self._modules[str(i)] = self._modules[str(i - 1)]
        self._modules[str(index)] = module

    def append(self, module):
        self._modules[str(len(self._modules))] = module
        return self

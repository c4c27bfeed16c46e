"""Sequential: torch's container with the reference's three builder methods and the spatial-size back-propagation every
SparseConvNet layer takes part in (reference: SparseConvNet/sparseconvnet/sequential.py:9-33).  Children are keyed
"0", "1", ... so state_dict keys match the reference's."""
import torch


class Sequential(torch.nn.Sequential):
    def input_spatial_size(self, out_size):
        """the input size this chain needs to produce `out_size`: asked of every layer, last to first"""
        for layer in reversed(list(self._modules.values())):
            out_size = layer.input_spatial_size(out_size)
        return out_size

    def _renumber(self, layers):
        self._modules.clear()
        for i, layer in enumerate(layers):
            self._modules[str(i)] = layer

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    append = add

    def insert(self, index, module):
        layers = list(self._modules.values())
        layers.insert(index, module)
        self._renumber(layers)

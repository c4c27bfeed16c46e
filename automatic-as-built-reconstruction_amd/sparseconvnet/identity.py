"""Identity (reference: SparseConvNet/sparseconvnet/identity.py:10-15): the pass-through branch of a ConcatTable."""
from torch.nn import Module


class Identity(Module):
    @staticmethod
    def input_spatial_size(out_size):
        """a grid of any size passes through unchanged"""
        return out_size

    def forward(self, input):
        return input

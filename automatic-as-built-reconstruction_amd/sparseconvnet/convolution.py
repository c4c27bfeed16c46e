"""Convolution: strided sparse convolution onto the coarser grid (reference:
SparseConvNet/sparseconvnet/convolution.py:14-90; layer machinery shared in _sparseConv.py)."""
from ._sparseConv import SparseConvModule


class Convolution(SparseConvModule):
    kind = "conv"

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        self._setup(dimension, nIn, nOut, filter_size, filter_stride, bias, groups)

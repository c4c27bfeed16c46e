"""Deconvolution: the transpose of `Convolution` back onto the finer grid stored in the Metadata (reference:
SparseConvNet/sparseconvnet/deconvolution.py:13-87; layer machinery shared in _sparseConv.py)."""
from ._sparseConv import SparseConvModule


class Deconvolution(SparseConvModule):
    kind = "deconv"

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        self._setup(dimension, nIn, nOut, filter_size, filter_stride, bias, groups)

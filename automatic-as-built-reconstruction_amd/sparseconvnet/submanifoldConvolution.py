"""SubmanifoldConvolution / ValidConvolution: output sites = input sites (reference:
SparseConvNet/sparseconvnet/submanifoldConvolution.py:14-113; layer machinery shared in _sparseConv.py)."""
from ._sparseConv import SparseConvModule


class SubmanifoldConvolution(SparseConvModule):
    kind = "subm"

    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        self._setup(dimension, nIn, nOut, filter_size, None, bias, groups)


class ValidConvolution(SubmanifoldConvolution):
    pass

"""Helpers of the operator API (reference: SparseConvNet/sparseconvnet/utils.py:10-66)."""
import torch

from .sparseConvNetTensor import SparseConvNetTensor
from .metadata import Metadata


def toLongTensor(dimension, x):
    if hasattr(x, "type") and x.type() == "torch.LongTensor":
        return x
    elif isinstance(x, (list, tuple)):
        assert len(x) == dimension
        return torch.LongTensor(list(x))
    else:
        return torch.LongTensor(dimension).fill_(x)


def optionalTensor(a, b):
    return getattr(a, b) if hasattr(a, b) else torch.Tensor()


def optionalTensorReturn(a):
    return a if a.numel() else None


def concatenate_feature_planes(input):
    output = SparseConvNetTensor()
    output.metadata = input[0].metadata
    output.spatial_size = input[0].spatial_size
    output.features = torch.cat([i.features for i in input], 1)
    return output


def add_feature_planes(input):
    output = SparseConvNetTensor()
    output.metadata = input[0].metadata
    output.spatial_size = input[0].spatial_size
    output.features = _sum_features(input)
    return output


def _sum_features(tensors):
    """left-to-right sum like the reference's `sum([...])`, minus its leading `0 +` (one
    elementwise launch saved; x + 0 is exact, so the result is bit-identical)"""
    f = tensors[0].features
    for t in tensors[1:]:
        f = f + t.features
    return f

"""Metadata factory (reference: SparseConvNet/sparseconvnet/metadata.py:16-17)."""
from . import SCN


def Metadata(dim):
    if dim != 3:
        raise NotImplementedError("the MI355X hot path implements dimension 3 (Metadata_3) only")
    return SCN.Metadata_3()

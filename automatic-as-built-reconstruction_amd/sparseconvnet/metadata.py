"""Metadata factory (reference: SparseConvNet/sparseconvnet/metadata.py:16-17)."""
from . import SCN


def Metadata(dim, site_order="first_seen"):
    """`site_order` (extension, see SCN.Metadata_3): "first_seen" = the reference's numbering, "brick" = brick-major"""
    if dim != 3:
        raise NotImplementedError("the MI355X hot path implements dimension 3 (Metadata_3) only")
    return SCN.Metadata_3(site_order)

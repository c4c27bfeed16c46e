"""reference: maskrcnn_benchmark/layers/__init__.py (the 3-D path imports `nms` only,
structures/boxlist_ops_3d.py:7)."""
from .nms import nms

__all__ = ["nms"]

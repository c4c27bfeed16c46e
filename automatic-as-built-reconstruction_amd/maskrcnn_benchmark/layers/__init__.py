"""reference: maskrcnn_benchmark/layers/__init__.py (the 3-D path imports `nms` only,
structures/boxlist_ops_3d.py:7)."""
from .nms import nms
from .roi_align_rotated_3d import ROIAlignRotated3D, roi_align_rotated_3d

__all__ = ["nms", "ROIAlignRotated3D", "roi_align_rotated_3d"]

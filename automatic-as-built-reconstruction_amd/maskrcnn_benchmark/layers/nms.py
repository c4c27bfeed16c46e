"""reference: maskrcnn_benchmark/layers/nms.py:5-7"""
from . import _C

nms = _C.nms

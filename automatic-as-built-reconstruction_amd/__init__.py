"""MI355X-native hot path of xuyongzhi/Automatic-As-built-Reconstruction.

This directory is laid out so that putting it FIRST on ``sys.path`` makes the reference's
own import statements resolve to this implementation::

    import sparseconvnet as scn                      # operator API (SparseConvNet/sparseconvnet)
    from maskrcnn_benchmark.layers import nms        # `_C.nms` surface
    from second.pytorch.core.box_torch_ops import rotate_nms_3d
    from utils3d.rotate_nms_3d_torch import boxes_iou_3d

``importlib.import_module("automatic-as-built-reconstruction_amd")`` performs that path insertion.
The compute path is ``lib/libaabr_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/aabr_hip.h``); there is no CPU fallback: using an operator without the library or
without a GPU raises.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "lib", "libaabr_hip.so")

if PKG_DIR not in sys.path:
    sys.path.insert(0, PKG_DIR)


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libaabr_hip.so (hipcc cross-compiles)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(PKG_DIR, "csrc"), "-j4"], stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce %s" % LIB_PATH)
    return LIB_PATH

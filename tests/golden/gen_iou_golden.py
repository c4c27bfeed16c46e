"""Generate tests/golden/iou_golden.npz by EXECUTING the reference's own rotated-IoU
device-function bodies (second/core/non_max_suppression/nms_gpu.py) on the host.

Runs only in the build container (needs /root/reference).  numba, numba.cuda and
spconv are not installed; the reference's device functions are plain-Python
bodies under `@cuda.jit(device=True)` decorators, so this script registers
decorator-only placeholder modules (identity decorators, `cuda.local.array` ->
numpy float32 zeros) purely so that the module can be imported; every arithmetic
statement that runs is the reference's.  The kernel launch wrappers cannot run
without CUDA, so the pair loop of rotate_iou_kernel_eval (:626-664: query box
first, then box) and check_same_boxes (:706-717, called as-is) are driven here.

Known limitation: float32 numpy scalars stand in for device
fp32 and math.cos/sin/sqrt run in double, so the vectors carry ~1e-6 tolerance.
The committed fixture is data only (inputs + expected outputs).
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth_scenes  # noqa: E402


def _install_placeholders():
    def deco(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba = types.ModuleType("numba")
    numba.jit = deco
    numba.njit = deco
    numba.float32 = np.float32
    numba.int32 = np.int32
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = deco

    class _Local:
        @staticmethod
        def array(shape, dtype=np.float32):
            return np.zeros(shape, np.float32)

    cuda.local = _Local
    cuda.shared = _Local
    numba.cuda = cuda
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cuda
    spconv = types.ModuleType("spconv")
    sputils = types.ModuleType("spconv.utils")
    for n in ("non_max_suppression", "non_max_suppression_cpu", "rotate_non_max_suppression_cpu", "rbbox_iou",
              "rbbox_intersection"):
        setattr(sputils, n, None)
    spconv.utils = sputils
    sys.modules["spconv"] = spconv
    sys.modules["spconv.utils"] = sputils


def ref_iou_matrix(mod, boxes, query, criterion):
    N, K = boxes.shape[0], query.shape[0]
    iou = np.zeros((N, K), np.float32)
    for n in range(N):
        for k in range(K):
            iou[n, k] = np.float32(mod.devRotateIoUEval(query[k], boxes[n], criterion))
    raw = iou.copy()
    mod.check_same_boxes(iou, boxes, query)
    return raw, iou


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    mod = importlib.import_module("second.core.non_max_suppression.nms_gpu")
    out = {}
    # (1) the boxes of the reference's own second/core/non_max_suppression/test_nms_gpu.py:6-11
    tb = np.array([[0, 0, 1, 2., 0.1], [0, 0, .001, 2., 0.1], [0, 0, 0.1, 2., 0.5], [0, 0, 0.1, 2., -np.pi / 2]],
                  np.float32)
    out["t_boxes"] = tb
    out["t_raw"], out["t_iou"] = ref_iou_matrix(mod, tb, tb, -1)
    # (2) hand-written wall boxes of maskrcnn_benchmark/structures/boxlist_ops_3d.py:124-175 style: thin walls,
    #     yaw 0 / pi/2, touching and crossing configurations
    hw = np.array([[2.0, 3.0, 0.1, 4.0, 0.0], [2.0, 3.0, 0.1, 4.0, np.pi / 2], [2.05, 3.0, 0.1, 4.0, 0.0],
                   [2.0, 3.5, 0.12, 3.0, 0.02], [6.0, 1.0, 0.3, 2.0, -1.2], [6.1, 1.1, 0.28, 2.2, -1.15],
                   [9.0, 9.0, 0.2, 0.5, 0.7]], np.float32)
    out["hw_boxes"] = hw
    for c in (-1, 0, 1, 2, 6):
        out["hw_raw_c%d" % c], out["hw_iou_c%d" % c] = ref_iou_matrix(mod, hw, hw, c)
    # (3) seeded clustered NMS set (synth_scenes.make_nms_boxes), first 96 boxes, 2-D columns [0,1,3,4,6]
    b7, sc = synth_scenes.make_nms_boxes(2000, 0)
    b5 = np.ascontiguousarray(b7[:96][:, [0, 1, 3, 4, 6]])
    out["nms_boxes7"] = b7[:96]
    out["nms_scores"] = sc[:96]
    out["nms_raw"], out["nms_iou"] = ref_iou_matrix(mod, b5, b5, -1)
    # (4) GT x anchors labelling shape, criterion 6 and -1 (rpn/loss_3d.py:95)
    g5 = np.ascontiguousarray(b7[100:108][:, [0, 1, 3, 4, 6]])
    a5 = np.ascontiguousarray(b7[200:264][:, [0, 1, 3, 4, 6]])
    out["lab_gt"], out["lab_anchor"] = g5, a5
    for c in (-1, 6):
        out["lab_raw_c%d" % c], out["lab_iou_c%d" % c] = ref_iou_matrix(mod, g5, a5, c)
    np.savez_compressed(os.path.join(HERE, "iou_golden.npz"), **out)
    print("wrote iou_golden.npz:", {k: v.shape for k, v in out.items()})
    print("test_nms_gpu boxes, raw diag:", np.diag(out["t_raw"]), "after check_same_boxes:", np.diag(out["t_iou"]))


if __name__ == "__main__":
    main()

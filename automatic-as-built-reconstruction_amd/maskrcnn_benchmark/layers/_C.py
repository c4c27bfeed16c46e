"""`_C` -- the maskrcnn_benchmark native-extension surface used by the 3-D detector
(reference: maskrcnn_benchmark/csrc/vision.cpp:9-20, imported as top-level `_C` through
maskrcnn_benchmark/layers/__init__.py:5-6)."""
import torch

import _nms


def nms(dets, scores, threshold):
    """csrc/nms.h:10-28.  Axis-aligned NMS with the +1 pixel convention; returns the kept
    indices in ascending index order (cpu/nms_cpu.cpp:66 `nonzero(suppressed == 0)`;
    cuda/nms.cu:125-130 sorts them as well)."""
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.long, device=dets.device)
    order = torch.sort(scores, 0, descending=True)[1]
    keep_sorted = _nms.nms_sorted(dets[order], threshold)
    return torch.sort(order[keep_sorted])[0]


def roi_align_rotated_3d_forward(*args):
    raise NotImplementedError("ROIAlignRotated3D is SURVEY.md §8(f) rank 3 (next), not part of the hot path yet")


def roi_align_rotated_3d_backward(*args):
    raise NotImplementedError("ROIAlignRotated3D is SURVEY.md §8(f) rank 3 (next), not part of the hot path yet")

"""rotate_nms_3d_cc (reference: second/core/non_max_suppression/nms_cpu.py:32-44).

The reference computes the IoU matrix on the GPU, copies it to the host and finishes in the
third-party `spconv.utils.rotate_non_max_suppression_cpu`; here the whole thing is one
device-side mask + scan (no host round trip).  Suppression rule: DESIGN.md §NMS."""
import torch

import _nms


def rotate_nms_3d_cc(dets, thresh, flag):
    assert dets.shape[1] == 8
    scores = dets[:, -1]
    order = torch.sort(scores, descending=True, stable=True)[1]
    keep_sorted = _nms.rotate_nms_sorted(dets[order, 0:7], thresh, -1, _nms.REFERENCE_DEBUG_ONLY_XY)
    return order.to(keep_sorted.device)[keep_sorted]

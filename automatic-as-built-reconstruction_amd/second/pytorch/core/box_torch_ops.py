"""rotate_nms_3d (reference: second/pytorch/core/box_torch_ops.py:557-582)."""
import torch

import _nms


def rotate_nms_3d(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5, flag=""):
    indices = None
    if pre_max_size is not None:
        num_keeped_scores = scores.shape[0]
        pre_max_size = min(num_keeped_scores, pre_max_size)
        scores, indices = torch.topk(scores, k=pre_max_size)  # sorted, descending
        rbboxes = rbboxes[indices]
    else:
        order = torch.sort(scores, descending=True, stable=True)[1]
    if rbboxes.shape[0] == 0:
        return torch.zeros([0]).long().to(rbboxes.device)
    if indices is not None:
        keep = _nms.rotate_nms_sorted(rbboxes, iou_threshold, -1 if post_max_size is None else post_max_size,
                                      _nms.REFERENCE_DEBUG_ONLY_XY)
        return indices.to(keep.device)[keep]
    keep = _nms.rotate_nms_sorted(rbboxes[order], iou_threshold, -1 if post_max_size is None else post_max_size,
                                  _nms.REFERENCE_DEBUG_ONLY_XY)
    return order.to(keep.device)[keep]

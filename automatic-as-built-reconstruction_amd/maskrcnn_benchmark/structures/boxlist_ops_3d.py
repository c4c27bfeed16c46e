"""NMS / IoU entry points of the detector (reference:
maskrcnn_benchmark/structures/boxlist_ops_3d.py:14-62,82-89).  `boxlist` is duck-typed: it
needs `.bbox3d` ([n,7] yx_zb), `.get_field(name)`, `len()`, `.mode` and `__getitem__` with a
LongTensor, which is what the reference's BoxList3D offers."""
import torch

from maskrcnn_benchmark.layers import nms as _box_nms  # noqa: F401  (import parity, :7)
from second.pytorch.core.box_torch_ops import rotate_nms_3d
from utils3d.rotate_nms_3d_torch import boxes_iou_3d


def boxlist_nms_3d(boxlist, nms_thresh, nms_aug_thickness=None, max_proposals=-1, score_field="score", flag=""):
    if nms_aug_thickness is None:
        nms_aug_thickness = [0, 0]
    if flag == "rpn_post":
        assert max_proposals > 100, max_proposals
    elif flag == "roi_post":
        assert max_proposals == -1
    else:
        raise NotImplementedError
    if max_proposals < 0:
        max_proposals = 500
    objectness = boxlist.get_field(score_field)
    bbox3d = boxlist.bbox3d.clone().detach()
    bbox3d[:, 3:5] = torch.clamp(bbox3d[:, 3:5], min=nms_aug_thickness[0])
    bbox3d[:, 5] = torch.clamp(bbox3d[:, 5], min=nms_aug_thickness[1])
    keep = rotate_nms_3d(bbox3d, objectness, pre_max_size=2000, post_max_size=max_proposals,
                         iou_threshold=nms_thresh, flag=flag)
    return boxlist[keep]


def boxlist_iou_3d(targets, anchors, aug_thickness, criterion, only_xy=False, flag=""):
    assert targets.mode == "yx_zb"
    assert anchors.mode == "yx_zb"
    return boxes_iou_3d(targets.bbox3d, anchors.bbox3d, aug_thickness, criterion, only_xy, flag)

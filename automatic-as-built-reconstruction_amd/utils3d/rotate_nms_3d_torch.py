"""boxes_iou_3d / iou_one_dim (reference: utils3d/rotate_nms_3d_torch.py:5-90)."""
import torch

import _nms

DEBUG = 1  # the reference module constant: forces only_xy (:5,32-33)


def iou_one_dim(targets_z, anchors_z):
    anchors_z = anchors_z.clone()
    targets_z = targets_z.clone()
    anchors_z[:, 1] = anchors_z[:, 0] + anchors_z[:, 1]
    targets_z[:, 1] = targets_z[:, 0] + targets_z[:, 1]
    targets_z = targets_z.unsqueeze(1)
    anchors_z = anchors_z.unsqueeze(0)
    overlap = torch.min(anchors_z[:, :, 1], targets_z[:, :, 1]) - torch.max(anchors_z[:, :, 0], targets_z[:, :, 0])
    common = torch.max(anchors_z[:, :, 1], targets_z[:, :, 1]) - torch.min(anchors_z[:, :, 0], targets_z[:, :, 0])
    return overlap / common


def boxes_iou_3d(targets_bbox3d, anchors_bbox3d, aug_thickness=None, criterion=-1, only_xy=False, flag=""):
    if DEBUG:
        only_xy = True
    if flag == "rpn_label_generation":
        assert aug_thickness["anchor_Y"] == 0
        assert aug_thickness["target_Y"] >= 0.3
    elif flag == "roi_label_generation":
        assert aug_thickness["anchor_Y"] >= 0.3
        assert aug_thickness["target_Y"] >= 0.3
    elif flag == "eval":
        assert aug_thickness["anchor_Y"] <= 0.3
        assert aug_thickness["target_Y"] <= 0.3
    elif flag == "rpn_post" or flag == "roi_post":
        assert aug_thickness is None
    else:
        raise NotImplementedError(flag)
    if aug_thickness is None:
        aug_thickness = {"target_Y": 0.0, "target_Z": 0.0, "anchor_Y": 0.0, "anchor_Z": 0.0}
    aug = (aug_thickness["target_Y"], aug_thickness["target_Z"], aug_thickness["anchor_Y"],
           aug_thickness["anchor_Z"])
    iou = _nms.boxes_iou_3d(targets_bbox3d.detach(), anchors_bbox3d.detach(), aug, criterion, only_xy)
    return iou.to(targets_bbox3d.device) if targets_bbox3d.is_cuda else iou

"""rotate_iou_gpu_eval (reference: second/core/non_max_suppression/nms_gpu.py:667-717): numpy
in, numpy out, same argument order; the pair loop, the rotated-rectangle clipping and the
`check_same_boxes` patch run in one HIP kernel."""
import numpy as np
import torch

import _nms


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    box_dtype = boxes.dtype
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float32)
    dev = torch.device("cuda", device_id if device_id is not None else 0)
    iou = _nms.rotate_iou_eval(torch.from_numpy(boxes).to(dev), torch.from_numpy(query_boxes).to(dev), criterion)
    return iou.cpu().numpy().astype(box_dtype)

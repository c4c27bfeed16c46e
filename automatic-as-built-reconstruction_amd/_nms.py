"""Device-resident rotated IoU / NMS on the C ABI (include/aabr_hip.h) -- shared by the
reference-named wrappers in second/, utils3d/ and maskrcnn_benchmark/."""
import torch

import _hip
from _hip import ptr, stream, check

# utils3d/rotate_nms_3d_torch.py:5 sets a module-level DEBUG = 1 that forces only_xy=True (:32-33)
REFERENCE_DEBUG_ONLY_XY = True


def _dev(t, device=None):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    _hip.require_gpu()
    return t.to(device=device, dtype=torch.float32).contiguous()


def rotate_iou_eval(boxes, query, criterion=-1):
    """[N,5] x [K,5] -> [N,K] float32 on the device (rotate_iou_gpu_eval semantics)."""
    boxes = _dev(boxes)
    query = _dev(query, boxes.device)
    N, K = boxes.size(0), query.size(0)
    iou = torch.zeros((N, K), dtype=torch.float32, device=boxes.device)
    check(_hip.load().aabr_rotate_iou_eval(ptr(boxes), N, ptr(query), K, int(criterion), ptr(iou), stream()))
    return iou


def boxes_iou_3d(targets, anchors, aug=(0.0, 0.0, 0.0, 0.0), criterion=-1, only_xy=True):
    targets = _dev(targets)
    anchors = _dev(anchors, targets.device)
    M, K = targets.size(0), anchors.size(0)
    # every entry is written by the kernel; only an empty side leaves the matrix untouched (and it is empty then)
    iou = torch.empty((M, K), dtype=torch.float32, device=targets.device)
    check(_hip.load().aabr_boxes_iou_3d(ptr(targets), M, ptr(anchors), K, _hip.f32x4(aug), int(criterion),
                                        int(bool(only_xy)), ptr(iou), stream()))
    return iou


def rotate_nms_sorted(boxes7_sorted, thresh, post_max=-1, only_xy=True, lazy=False):
    """boxes already in descending-score order -> LongTensor of kept positions (device).
    `lazy`: no host read here -- returns (keep [n], meta) with the number kept in meta[0] on the device, so that a
    caller with several lists to suppress can enqueue them all and read the counts once."""
    b = _dev(boxes7_sorted)
    n = b.size(0)
    if n == 0:
        z = torch.zeros(0, dtype=torch.int64, device=b.device)
        return (z, None) if lazy else z
    cb = (n + 63) // 64
    mask = torch.empty(n * cb, dtype=torch.int64, device=b.device)
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    meta = torch.empty(_hip.META_WORDS, dtype=torch.int32, device=b.device)
    check(_hip.load().aabr_rotate_nms_sorted(ptr(b), n, float(thresh), int(bool(only_xy)),
                                             int(post_max if post_max is not None else -1), ptr(mask), ptr(keep),
                                             ptr(meta), stream()))
    if lazy:
        return keep, meta
    nk = int(meta[0].item())
    return keep[:nk]


def nms_sorted(dets4_sorted, thresh):
    b = _dev(dets4_sorted)
    n = b.size(0)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=b.device)
    cb = (n + 63) // 64
    mask = torch.empty(n * cb, dtype=torch.int64, device=b.device)
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    meta = torch.empty(_hip.META_WORDS, dtype=torch.int32, device=b.device)
    check(_hip.load().aabr_nms_sorted(ptr(b), n, float(thresh), ptr(mask), ptr(keep), ptr(meta), stream()))
    return keep[: int(meta[0].item())]

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


def roi_align_rotated_3d_forward(input, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize,
                                 sampling_ratio):
    """csrc/vision.cpp:19, csrc/cuda/ROIAlignRotated3D_cuda.cu:349-398: input [B,C,H,W,Z], rois [n,8] ->
    [n, C, ph, pw, pz]"""
    import _hip
    from _hip import ptr, stream, check
    _hip.require_gpu(input)
    inp = input.contiguous().float()
    r = rois.contiguous().float()
    B, Cc, H, W, Z = inp.shape
    out = torch.empty((r.size(0), Cc, int(pooled_height), int(pooled_width), int(pooled_zsize)), dtype=torch.float32,
                      device=inp.device)
    check(_hip.load().aabr_roi_align_rotated_3d_forward(ptr(inp), ptr(r), r.size(0), float(spatial_scale), Cc, H, W,
                                                        Z, int(pooled_height), int(pooled_width), int(pooled_zsize),
                                                        int(sampling_ratio), ptr(out), stream()))
    return out


def roi_align_rotated_3d_backward(grad, rois, spatial_scale, pooled_height, pooled_width, pooled_zsize, batch_size,
                                  channels, height, width, zsize, sampling_ratio):
    """csrc/vision.cpp:20, ROIAlignRotated3D_cuda.cu:401-454"""
    import _hip
    from _hip import ptr, stream, check
    _hip.require_gpu(grad)
    g = grad.contiguous().float()
    r = rois.contiguous().float()
    gin = torch.empty((int(batch_size), int(channels), int(height), int(width), int(zsize)), dtype=torch.float32,
                      device=g.device)
    check(_hip.load().aabr_roi_align_rotated_3d_backward(ptr(g), ptr(r), r.size(0), float(spatial_scale),
                                                         int(pooled_height), int(pooled_width), int(pooled_zsize),
                                                         int(batch_size), int(channels), int(height), int(width),
                                                         int(zsize), int(sampling_ratio), ptr(gin), stream()))
    return gin

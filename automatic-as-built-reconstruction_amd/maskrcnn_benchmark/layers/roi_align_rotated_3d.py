"""ROIAlignRotated3D (reference: maskrcnn_benchmark/layers/roi_align_rotated_3d.py:11-94)."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from sparseconvnet.tools_3d_2d import sparse_3d_to_dense_2d
from . import _C


class _ROIAlignRotated3D(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(roi)
        ctx.output_size = tuple(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        return _C.roi_align_rotated_3d_forward(input, roi, spatial_scale, output_size[0], output_size[1],
                                               output_size[2], sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        output_size = ctx.output_size
        bs, ch, h, w, zsize = ctx.input_shape
        grad_input = _C.roi_align_rotated_3d_backward(grad_output, rois, ctx.spatial_scale, output_size[0],
                                                      output_size[1], output_size[2], bs, ch, h, w, zsize,
                                                      ctx.sampling_ratio)
        return grad_input, None, None, None, None


roi_align_rotated_3d = _ROIAlignRotated3D.apply


def _occupied_extent(input_s3d):
    """(x_size, y_size, z_size, batch_size) = max coordinate + 1 per column, as sparse_3d_to_dense_2d crops
    (tools_3d_2d.py:8-10); one 16-byte read-back per grid, cached on the metadata"""
    import torch
    md = input_s3d.metadata
    key = tuple(int(v) for v in input_s3d.spatial_size.tolist())
    cache = md.__dict__.setdefault("_roi_extent", {}) if hasattr(md, "__dict__") else {}
    ext = cache.get(key)
    if ext is None:
        g = md.grids[key]
        ext = tuple((g.coords.max(0)[0] + 1).tolist()) if g.V else (0, 0, 0, 0)
        cache[key] = ext
    return ext


def _cellmap(input_s3d, ext):
    """int32 [B, X, Y, Z] over the occupied extent: site row or -1; built once per grid"""
    import torch
    import _hip
    from _hip import ptr, stream, check
    md = input_s3d.metadata
    key = tuple(int(v) for v in input_s3d.spatial_size.tolist())
    cache = md.__dict__.setdefault("_roi_cellmap", {})
    cm = cache.get(key)
    if cm is None:
        g = md.grids[key]
        x, y, z, b = ext
        cm = torch.empty((b, x, y, z), dtype=torch.int32, device=g.coords.device)
        check(_hip.load().aabr_roi_cellmap(ptr(g.coords), g.V, _hip.i32x3((x, y, z)), b, ptr(cm), stream()))
        cache[key] = cm
    return cm


class _ROIAlignRotated3DSparse(Function):
    """densify + crop + ROI-align as ONE gather from the sparse feature rows (bit-identical forward)"""

    @staticmethod
    def forward(ctx, features, cellmap, roi, output_size, spatial_scale, sampling_ratio):
        import torch
        import _hip
        from _hip import ptr, stream, check
        feats = features.contiguous()
        rois = roi.to(device=feats.device, dtype=torch.float32).contiguous()
        b, x, y, z = cellmap.shape
        out = torch.empty((rois.size(0), feats.size(1)) + tuple(output_size), dtype=torch.float32, device=feats.device)
        check(_hip.load().aabr_roi_align_rotated_3d_sparse_forward(
            ptr(feats), feats.size(1), ptr(cellmap), b, x, y, z, ptr(rois), rois.size(0), float(spatial_scale),
            int(output_size[0]), int(output_size[1]), int(output_size[2]), int(sampling_ratio), ptr(out), stream()))
        ctx.save_for_backward(rois, cellmap)
        ctx.geom = (tuple(output_size), float(spatial_scale), int(sampling_ratio), feats.size(0), feats.size(1))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        import torch
        import _hip
        from _hip import ptr, stream, check
        rois, cellmap = ctx.saved_tensors
        output_size, scale, sampling, V, C = ctx.geom
        b, x, y, z = cellmap.shape
        g = grad_output.contiguous()
        d_feats = torch.empty((V, C), dtype=torch.float32, device=g.device)
        check(_hip.load().aabr_roi_align_rotated_3d_sparse_backward(
            ptr(g), C, ptr(cellmap), b, x, y, z, ptr(rois), rois.size(0), scale, output_size[0], output_size[1],
            output_size[2], sampling, V, ptr(d_feats), stream()))
        return d_feats, None, None, None, None, None


class ROIAlignRotated3D(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlignRotated3D, self).__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.fused = True  # False: the reference's two steps (dense tensor, then `_C.roi_align_rotated_3d_*`)

    def forward(self, input_s3d, rois_3d):
        import torch
        if self.fused and input_s3d.features.dtype == torch.float32 and input_s3d.features.dim() == 2 \
                and input_s3d.features.size(0) > 0:
            ext = _occupied_extent(input_s3d)
            return _ROIAlignRotated3DSparse.apply(input_s3d.features, _cellmap(input_s3d, ext), rois_3d,
                                                  self.output_size, self.spatial_scale, self.sampling_ratio)
        input_d3d = sparse_3d_to_dense_2d(input_s3d)
        return roi_align_rotated_3d(input_d3d, rois_3d, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale=" +
                str(self.spatial_scale) + ", sampling_ratio=" + str(self.sampling_ratio) + ")")

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


class ROIAlignRotated3D(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlignRotated3D, self).__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input_s3d, rois_3d):
        input_d3d = sparse_3d_to_dense_2d(input_s3d)
        return roi_align_rotated_3d(input_d3d, rois_3d, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale=" +
                str(self.spatial_scale) + ", sampling_ratio=" + str(self.sampling_ratio) + ")")

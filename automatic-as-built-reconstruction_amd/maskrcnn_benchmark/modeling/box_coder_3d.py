"""BoxCoder3D -- regression-target encode / box decode of the RPN on the device (reference:
maskrcnn_benchmark/modeling/box_coder_3d.py:12-80, centroid form; the corner form, `is_corner_roi`, belongs to the ROI
heads and is outside this path).  `encode` is what RPNLossComputation.prepare_targets calls per image
(modeling/rpn/loss_3d.py:186-196), `decode` what RPNPostProcessor calls on the selected anchors
(modeling/rpn/inference_3d.py:124-127); in the training step both are fused into the label / proposal kernels
(rpn_glue.rpn_label_matches(regression_targets=True), rpn_glue.rpn_proposals) -- this class is the list form."""
import torch

import _hip
from _hip import check, ptr, stream


class BoxCoder3D(object):
    def __init__(self, is_corner_roi, weights):
        if is_corner_roi:
            raise ValueError("the corner-box coder of the ROI heads is not part of the RPN path")
        self.is_corner_roi = False
        self.smooth_dim = True
        self.weights = torch.tensor((1.0,) * 7 if weights is None else weights, dtype=torch.float32).view(1, 7)
        self.bbox_xform_clip = 10000. / 1

    def _w(self):
        return _hip.f32xn(self.weights.view(7).tolist())

    def encode(self, targets, anchors):
        return self.encode_centroid_box(targets, anchors)

    def decode(self, box_encodings, anchors):
        return self.decode_centroid_box(box_encodings, anchors)

    def encode_centroid_box(self, targets, anchors):
        assert targets.shape == anchors.shape and anchors.shape[1] == 7
        t = targets.to(torch.float32).contiguous()
        a = anchors.to(device=t.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(t)
        check(_hip.load().aabr_box_encode(ptr(t), ptr(a), t.shape[0], self._w(), ptr(out), stream()))
        return out

    def decode_centroid_box(self, box_encodings, anchors):
        assert box_encodings.shape[0] == anchors.shape[0]
        assert anchors.shape[1] == 7
        num_classes = int(box_encodings.shape[1] / 7)
        e = box_encodings.to(torch.float32).contiguous()
        a = anchors.to(device=e.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(e)
        check(_hip.load().aabr_box_decode(ptr(e), ptr(a), e.shape[0], num_classes, self._w(),
                                          float(self.bbox_xform_clip), ptr(out), stream()))
        return out

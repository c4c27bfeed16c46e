"""Device-resident RPN glue: from a sparse feature map's site list and the RPN
head outputs to NMS-ed proposals without a host round trip.

Restates, for one feature map, `RPNPostProcessor.forward_for_single_feature_map`
(maskrcnn_benchmark/modeling/rpn/inference_3d.py:82-163): per example sigmoid -> top-k ->
anchors of the selected indices (anchor_generator_sparse3d.py:88-104) -> BoxCoder3D.decode ->
boxlist_nms_3d (structures/boxlist_ops_3d.py:14-62).  sigmoid / top-k are torch plumbing; anchor
generation + decode are one fused HIP kernel; NMS is the device mask + scan.  Containers
(BoxList3D) are out of scope: plain tensors in, plain tensors out."""
import torch

import _hip
import _nms
from _hip import ptr, stream, check


def grid_anchors(site_coords, base_anchors, voxel_scale, stride):
    """all anchors of a map, flattened [site, yaw] like AnchorGenerator.grid_anchors: [V*A, 7]"""
    c = site_coords[:, 0:3].float() / voxel_scale * torch.as_tensor(stride, dtype=torch.float32,
                                                                    device=site_coords.device).view(1, 3)
    c = torch.cat([c, torch.zeros(c.shape[0], 4, device=c.device)], 1).view(-1, 1, 7)
    return (c + base_anchors.view(1, -1, 7).to(c.device)).reshape(-1, 7)


def rpn_proposals_single_map(tensor, objectness, box_regression, base_anchors, voxel_scale, stride,
                             pre_nms_top_n=2000, post_nms_top_n=1000, nms_thresh=0.5,
                             nms_aug_thickness=(0.3, 0.3), weights=(1.0,) * 7, bbox_xform_clip=10000.0):
    """tensor: SparseConvNetTensor of the map (sites batch-contiguous); objectness [V*A] logits and
    box_regression [V*A,7] in the flatten order [site, yaw].  Returns a list over examples of
    (boxes [m,7] yx_zb, objectness [m]) after NMS, all on the device."""
    lib = _hip.load()
    g = tensor.metadata.grids[tuple(int(v) for v in tensor.spatial_size.tolist())]
    dev = objectness.device
    A = int(base_anchors.shape[0])
    ba = base_anchors.to(device=dev, dtype=torch.float32).contiguous()
    reg = box_regression.contiguous().float()
    batch = g.coords[:, 3]
    nb = int(batch[-1].item()) + 1 if g.V else 0
    counts = torch.bincount(batch.long(), minlength=nb).tolist()   # sites per example (one small read-back)
    out, s = [], 0
    for bi in range(nb):
        e = s + counts[bi]
        n_anchor = (e - s) * A
        if n_anchor == 0:
            out.append((torch.zeros(0, 7, device=dev), torch.zeros(0, device=dev)))
            continue
        obj = objectness[s * A:e * A].sigmoid()
        k = min(pre_nms_top_n, n_anchor)
        score, idx = obj.topk(k, dim=0, sorted=True)
        boxes = torch.empty((k, 7), dtype=torch.float32, device=dev)
        check(lib.aabr_rpn_decode(ptr(g.coords), s, ptr(idx), k, ptr(reg), s * A, ptr(ba), A, float(voxel_scale),
                                  _hip.f32xn(stride), _hip.f32xn(weights), float(bbox_xform_clip), ptr(boxes),
                                  stream()))
        # boxlist_nms_3d: thickness clamps, then rotated NMS on the (already sorted) list
        nb7 = boxes.clone()
        nb7[:, 3:5] = torch.clamp(nb7[:, 3:5], min=nms_aug_thickness[0])
        nb7[:, 5] = torch.clamp(nb7[:, 5], min=nms_aug_thickness[1])
        keep = _nms.rotate_nms_sorted(nb7, nms_thresh, post_nms_top_n, _nms.REFERENCE_DEBUG_ONLY_XY)
        out.append((boxes[keep], score[keep]))
        s = e
    return out


def cat_scales_obj_reg(objectness, rpn_box_regression, examples_idxscope):
    """`cat_scales_obj_reg` (modeling/rpn/rpn_sparse3d.py:19-77) on plain tensors: the RPN head emits, per
    scale, the objectness / regression rows of all examples back to back; the loss and the post-processor
    want them regrouped example-major ([example][scale][row]).  `objectness[s]` reshapes to [rows_s, sep],
    `rpn_box_regression[s]` to [rows_s, 7*sep]; `examples_idxscope[s][b] = (begin, end)` is the row range
    of example b at scale s (the reference keeps it on its anchor BoxList3D).  Stays on the device; the
    regrouping is two `torch.cat`s of views."""
    scale_num = len(objectness)
    assert scale_num == len(rpn_box_regression) == len(examples_idxscope)
    batch_size = len(examples_idxscope[0])
    obj_new = [[] for _ in range(batch_size)]
    reg_new = [[] for _ in range(batch_size)]
    for s in range(scale_num):
        assert objectness[s].shape[0] == 1 and rpn_box_regression[s].shape[0] == 1
        sep = objectness[s].shape[-1]
        assert rpn_box_regression[s].shape[-1] == 7 * sep
        obj_s = objectness[s].reshape(-1, sep)
        reg_s = rpn_box_regression[s].reshape(-1, 7 * sep)
        for b in range(batch_size):
            begin, end = examples_idxscope[s][b]
            obj_new[b].append(obj_s[begin:end])
            reg_new[b].append(reg_s[begin:end])
    obj = torch.cat([torch.cat(o, 0) for o in obj_new], 0)
    reg = torch.cat([torch.cat(r, 0) for r in reg_new], 0)
    return obj, reg

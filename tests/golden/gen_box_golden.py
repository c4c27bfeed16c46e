"""Generate tests/golden/box_golden.npz by IMPORTING the reference's pure-torch box code and running it
on seeded inputs (build container only; needs /root/reference):

  * maskrcnn_benchmark/modeling/box_coder_3d.py:46-80   BoxCoder3D.encode/decode_centroid_box
    -> second/pytorch/core/box_torch_ops.py:82-154      second_box_encode / second_box_decode
    -> utils3d/geometric_torch.py:4-10                  limit_period
  * utils3d/rotate_nms_3d_torch.py:7-90                 boxes_iou_3d (+ iou_one_dim): thickness / height
    clamps per flag, column selection [0,1,3,4,6], module DEBUG=1 => only_xy
  * second/pytorch/core/box_torch_ops.py:557-582        rotate_nms_3d's top-k / index-remap shell
    (the suppression loop inside it is spconv's and cannot run: its stand-in here is the rule
    DESIGN.md section 4 states -- the matrix as `> 0` pre-filter, an exact polygon IoU of the reference's own
    corners for `>= thresh` -- so those vectors pin the SHELL and the corner convention, not the loop)

numba / spconv are absent: the decorator-only placeholder modules of gen_iou_golden.py let the modules
import; `rotate_iou_gpu_eval` (a CUDA launch wrapper) is replaced in the importing module's namespace
by the host pair-loop over the reference's own devRotateIoUEval + check_same_boxes, exactly as
gen_iou_golden.py drives it.  The committed fixture is data only (inputs + expected outputs).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_iou_golden as G  # noqa: E402
import synth_scenes  # noqa: E402

REF = "/root/reference"


def main():
    G._install_placeholders()
    import collections
    import collections.abc
    # harness-side py>=3.10 accommodation: torchplus/train/optim.py:1 does `from collections import Iterable`
    collections.Iterable = collections.abc.Iterable
    sys.path.insert(0, REF)
    nms_gpu = importlib.import_module("second.core.non_max_suppression.nms_gpu")
    bto = importlib.import_module("second.pytorch.core.box_torch_ops")
    rn3 = importlib.import_module("utils3d.rotate_nms_3d_torch")
    coder_mod = importlib.import_module("maskrcnn_benchmark.modeling.box_coder_3d")

    def host_rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
        # rotate_iou_gpu_eval (nms_gpu.py:667-703) without the CUDA launch
        boxes = boxes.astype(np.float32)
        query_boxes = query_boxes.astype(np.float32)
        _, iou = G.ref_iou_matrix(nms_gpu, boxes, query_boxes, criterion)
        return iou

    rn3.rotate_iou_gpu_eval = host_rotate_iou_gpu_eval
    out = {}
    rng = np.random.default_rng(7)

    # ---------------------------------------------------------------- BoxCoder3D centroid decode / encode
    b7, _ = synth_scenes.make_nms_boxes(600, 3)
    anchors = torch.from_numpy(b7[:256].copy())
    # wall-anchor-like spread incl. yaw around +-pi/2 so limit_period wraps
    anchors[:, 6] = torch.from_numpy(rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi / 4], 256).astype(np.float32))
    enc = torch.from_numpy((rng.standard_normal((256, 7)) * np.array([0.3, 0.3, 0.3, 0.4, 0.4, 0.2, 1.5]))
                           .astype(np.float32))
    enc[5, 3:6] = 20000.0   # beyond bbox_xform_clip (smooth_dim: 10000)
    out["dec_anchors"], out["dec_enc"] = anchors.numpy().copy(), enc.numpy().copy()
    for name, w in (("w1", None), ("w2", (10.0, 10.0, 10.0, 5.0, 5.0, 5.0, 2.0))):
        coder = coder_mod.BoxCoder3D(False, w)
        out["dec_%s" % name] = coder.decode(enc.clone(), anchors.clone()).numpy()
        out["weights_%s" % name] = coder.weights.numpy().reshape(7)
        tgt = torch.from_numpy(b7[300:556].copy())
        out["enc_targets"] = tgt.numpy().copy()
        out["enc_%s" % name] = coder.encode(tgt.clone(), anchors.clone()).numpy()
    # multi-class layout: [n, 7*num_classes] (box_coder_3d.py:64-69,77-78)
    enc3 = torch.from_numpy((rng.standard_normal((64, 21)) * 0.3).astype(np.float32))
    out["dec3_enc"] = enc3.numpy().copy()
    out["dec3_w1"] = coder_mod.BoxCoder3D(False, None).decode(enc3.clone(), anchors[:64].clone()).numpy()

    # ---------------------------------------------------------------- boxes_iou_3d clamps per flag
    t7 = torch.from_numpy(b7[100:112].copy())
    a7 = torch.from_numpy(b7[200:280].copy())
    t7[:, 3] = torch.from_numpy(rng.uniform(0.05, 0.5, 12).astype(np.float32))    # straddle the 0.3 clamp
    a7[:, 3] = torch.from_numpy(rng.uniform(0.05, 0.5, 80).astype(np.float32))
    t7[:, 5] = torch.from_numpy(rng.uniform(0.1, 2.8, 12).astype(np.float32))
    out["iou3d_targets"], out["iou3d_anchors"] = t7.numpy().copy(), a7.numpy().copy()
    cases = {
        "rpn_label_generation": ({"target_Y": 0.3, "target_Z": 0.3, "anchor_Y": 0.0, "anchor_Z": 0.0}, 6),
        "roi_label_generation": ({"target_Y": 0.3, "target_Z": 0.3, "anchor_Y": 0.3, "anchor_Z": 0.3}, -1),
        "eval": ({"target_Y": 0.2, "target_Z": 0.2, "anchor_Y": 0.2, "anchor_Z": 0.2}, -1),
        "rpn_post": (None, -1),
    }
    for flag, (aug, crit) in cases.items():
        iou = rn3.boxes_iou_3d(t7, a7, aug_thickness=aug, criterion=crit, only_xy=False, flag=flag)
        out["iou3d_%s" % flag] = iou.numpy()
        out["iou3d_%s_aug" % flag] = np.array([0, 0, 0, 0] if aug is None else
                                              [aug["target_Y"], aug["target_Z"], aug["anchor_Y"], aug["anchor_Z"]],
                                              np.float32)
        out["iou3d_%s_crit" % flag] = np.array(crit)
    assert rn3.DEBUG == 1   # => every case above is only_xy (rotate_nms_3d_torch.py:5,32-33)

    # ---------------------------------------------------------------- rotate_nms_3d shell
    nms_cpu = importlib.import_module("second.core.non_max_suppression.nms_cpu")

    def poly_iou(P, Q):
        # exact IoU of two convex quadrilaterals (float64 Sutherland-Hodgman + shoelace), written for this generator:
        # what a polygon library (boost::geometry in spconv 1.x) returns for intersection / union areas
        def area(p):
            return 0.5 * sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1]
                             for i in range(len(p)))
        P = [(float(x), float(y)) for x, y in P]
        Q = [(float(x), float(y)) for x, y in Q]
        sgn = 1.0 if area(Q) >= 0 else -1.0
        cur = P
        for e in range(4):
            a, b = Q[e], Q[(e + 1) % 4]
            nxt = []
            for i in range(len(cur)):
                p, q = cur[i], cur[(i + 1) % len(cur)]
                dp = sgn * ((b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]))
                dq = sgn * ((b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0]))
                if dp >= 0:
                    nxt.append(p)
                if (dp > 0 and dq < 0) or (dp < 0 and dq > 0):
                    t = dp / (dp - dq)
                    nxt.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
            cur = nxt
            if not cur:
                return 0.0
        inter = abs(area(cur)) if len(cur) >= 3 else 0.0
        return inter / (abs(area(P)) + abs(area(Q)) - inter)

    def stated_rule(corners, order, standup_iou, thresh):
        # stand-in for spconv.utils.rotate_non_max_suppression_cpu (un-vendored; its published 1.x loop): greedy over
        # `order`; j suppressed by kept i iff standup_iou[i, j] > 0 and the polygon IoU of corners[i], corners[j]
        # (the reference's own center_to_corner_box2d output) >= thresh
        n = len(order)
        sup = np.zeros(n, bool)
        keep = []
        for _i in range(n):
            i = order[_i]
            if sup[i]:
                continue
            keep.append(i)
            for _j in range(_i + 1, n):
                j = order[_j]
                if not sup[j] and standup_iou[i, j] > 0 and poly_iou(corners[i], corners[j]) >= thresh:
                    sup[j] = True
        return keep

    nms_cpu.rotate_non_max_suppression_cpu = stated_rule
    nb, ns = synth_scenes.make_nms_boxes(160, 5)
    out["nms3d_boxes"], out["nms3d_scores"] = nb, ns
    for pre, post in ((100, 30), (2000, 1000)):
        keep = bto.rotate_nms_3d(torch.from_numpy(nb), torch.from_numpy(ns), pre_max_size=pre, post_max_size=post,
                                 iou_threshold=0.5, flag="rpn_post")
        out["nms3d_keep_%d_%d" % (pre, post)] = keep.numpy().astype(np.int64)

    np.savez_compressed(os.path.join(HERE, "box_golden.npz"), **out)
    print("wrote box_golden.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

"""box_oracle.py -- numpy float32 restatement of the reference's RPN box glue.

TEST INFRASTRUCTURE ONLY (tests/, smoke, cpu_baseline): the product never imports this.
Pinned by tests/golden/box_golden.npz, which tests/golden/gen_box_golden.py produced by importing and
running the reference's own torch code (tests/test_oracle_golden.py::test_box_*).

Restates (citations under /root/reference):
  * BoxCoder3D.decode_centroid_box / encode_centroid_box  maskrcnn_benchmark/modeling/box_coder_3d.py:46-80
  * second_box_decode / second_box_encode (smooth_dim)     second/pytorch/core/box_torch_ops.py:82-154
  * limit_period                                           utils3d/geometric_torch.py:4-10
  * AnchorGenerator.grid_anchors                           maskrcnn_benchmark/modeling/rpn/anchor_generator_sparse3d.py:88-104
"""
import numpy as np

F = np.float32
PI = F(np.pi)   # torch evaluates `val / math.pi` with the python double rounded to the tensor dtype


def limit_period(val, offset, period):
    val = val.astype(F)
    return (val - np.floor(val / F(period) + F(offset)) * F(period)).astype(F)


def decode_centroid_box(box_encodings, anchors, weights=(1.0,) * 7, bbox_xform_clip=10000.0):
    enc = np.asarray(box_encodings, F)
    anchors = np.asarray(anchors, F)
    num_classes = enc.shape[1] // 7
    n = enc.shape[0]
    if num_classes != 1:
        enc = enc.reshape(-1, 7)
        anchors = np.repeat(anchors.reshape(n, 1, 7), num_classes, 1).reshape(-1, 7)
    e = (enc / np.asarray(weights, F).reshape(1, 7)).astype(F)
    e[:, 3:6] = np.minimum(e[:, 3:6], F(bbox_xform_clip))
    xa, ya, za, wa, la, ha, ra = [anchors[:, i] for i in range(7)]
    diag = np.sqrt((la * la + wa * wa).astype(F)).astype(F)
    out = np.stack([e[:, 0] * diag + xa, e[:, 1] * diag + ya, e[:, 2] * ha + za, (e[:, 3] + F(1)) * wa,
                    (e[:, 4] + F(1)) * la, (e[:, 5] + F(1)) * ha, e[:, 6] + ra], 1).astype(F)
    out[:, 6] = limit_period(out[:, 6], 0.5, PI)
    if num_classes != 1:
        out = out.reshape(n, num_classes * 7)
    return out


def encode_centroid_box(targets, anchors, weights=(1.0,) * 7):
    g = np.asarray(targets, F)
    a = np.asarray(anchors, F)
    xa, ya, za, wa, la, ha, ra = [a[:, i] for i in range(7)]
    xg, yg, zg, wg, lg, hg, rg = [g[:, i] for i in range(7)]
    diag = np.sqrt((la * la + wa * wa).astype(F)).astype(F)
    enc = np.stack([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / ha, wg / wa - F(1), lg / la - F(1),
                    hg / ha - F(1), rg - ra], 1).astype(F)
    enc[:, 6] = limit_period(enc[:, 6], 0.5, PI)
    return (enc * np.asarray(weights, F).reshape(1, 7)).astype(F)


def grid_anchors(site_coords, base_anchors, voxel_scale, stride):
    """[V*A, 7], flatten order [site, yaw]"""
    sc = np.asarray(site_coords)
    base = np.asarray(base_anchors, F)
    cen = (sc[:, :3].astype(F) / F(voxel_scale) * np.asarray(stride, F).reshape(1, 3)).astype(F)
    cen = np.concatenate([cen, np.zeros((sc.shape[0], 4), F)], 1)
    return (cen[:, None, :] + base[None]).reshape(-1, 7).astype(F)


def rpn_proposals(site_coords, objectness, box_regression, base_anchors, strides, voxel_scale, boxes_iou_3d,
                  nms_from_matrix, pre_nms_top_n=2000, post_nms_top_n=1000, nms_thresh=0.5,
                  nms_aug_thickness=(0.3, 0.3), weights=(1.0,) * 7, bbox_xform_clip=10000.0):
    """cat_scales_obj_reg (rpn_sparse3d.py:19-77) + RPNPostProcessor.forward_for_single_feature_map
    (rpn/inference_3d.py:95-149) + boxlist_nms_3d (structures/boxlist_ops_3d.py:14-62) on numpy arrays.
    site_coords[m] [V_m,4] (batch-contiguous), objectness[m] [V_m*A], box_regression[m] [V_m*A,7].
    `boxes_iou_3d`, `nms_from_matrix`: the C oracle's functions (tests/oracle_lib.py).
    Returns per example (boxes, scores, selected-local-indices)."""
    n_maps = len(site_coords)
    A = np.asarray(base_anchors[0]).shape[0]
    anchors = [grid_anchors(site_coords[m], base_anchors[m], voxel_scale, strides[m]) for m in range(n_maps)]
    nb = int(site_coords[0][:, 3].max()) + 1 if len(site_coords[0]) else 0
    out = []
    for bi in range(nb):
        o, r, a = [], [], []
        for m in range(n_maps):
            rows = np.nonzero(np.asarray(site_coords[m])[:, 3] == bi)[0]
            if len(rows) == 0:
                continue
            s, e = rows[0] * A, (rows[-1] + 1) * A
            o.append(np.asarray(objectness[m], F).reshape(-1)[s:e])
            r.append(np.asarray(box_regression[m], F).reshape(-1, 7)[s:e])
            a.append(anchors[m][s:e])
        if not o:
            out.append((np.zeros((0, 7), F), np.zeros(0, F), np.zeros(0, np.int64)))
            continue
        o, r, a = np.concatenate(o), np.concatenate(r), np.concatenate(a)
        score = (F(1) / (F(1) + np.exp(-o.astype(F)))).astype(F)
        k = min(pre_nms_top_n, len(o))
        idx = np.argsort(-o, kind="stable")[:k]          # sigmoid is monotone: same order as on the scores
        dec = decode_centroid_box(r[idx], a[idx], weights, bbox_xform_clip)
        nb7 = dec.copy()
        nb7[:, 3:5] = np.maximum(nb7[:, 3:5], F(nms_aug_thickness[0]))
        nb7[:, 5] = np.maximum(nb7[:, 5], F(nms_aug_thickness[1]))
        iou = boxes_iou_3d(nb7, nb7)
        keep = nms_from_matrix(iou, np.arange(k, dtype=np.int32), nms_thresh)[:post_nms_top_n]
        out.append((dec[keep], score[idx][keep], idx[keep]))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# RPN label generation: |yaw difference| + Matcher (pinned by tests/golden/matcher_golden.npz, which
# tests/golden/gen_matcher_golden.py produced by importing the reference's maskrcnn_benchmark/modeling/matcher.py and
# utils3d/geometric_torch.py; tests/test_oracle_golden.py::test_matcher_*)
def angle_dif(val0, val1):
    """utils3d/geometric_torch.py:12-21 with aim_scope_id 0: (val1 - val0) wrapped into [-pi/2, pi/2)"""
    return limit_period((np.asarray(val1, F) - np.asarray(val0, F)).astype(F), 0.5, PI)


BELOW_LOW_THRESHOLD, BETWEEN_THRESHOLDS = -1, -2


def matcher(match_quality_matrix, yaw_diff, high_threshold, low_threshold, allow_low_quality_matches=True,
            yaw_threshold=0.7):
    """Matcher.__call__ (maskrcnn_benchmark/modeling/matcher.py:57-100) as the RPN loss builds and calls it
    (rpn/loss_3d.py:91-100,338-344): yaw mask (:50-55), best ground truth per anchor with the first maximum
    (torch.max(dim=0) on the host), the two thresholds (:88-94), then set_low_quality_matches_ (:104-196) with the
    module flags as committed upstream (ENALE_SECOND_THIRD_MAX__ONLY_HIGHEST_IOU_TARGET False,
    IGNORE_HIGHEST_MATCH_NEARBY True, `cendis` unused: `if cendis is None or True`).
    match_quality_matrix [G, N] fp32, yaw_diff [G, N] fp32 or None -> (matches int64 [N], matched_vals fp32 [N])"""
    mq = np.asarray(match_quality_matrix, F)
    if yaw_diff is not None and not (yaw_threshold > 1.58):
        mq = (mq * (np.abs(np.asarray(yaw_diff, F)) < F(yaw_threshold)).astype(F)).astype(F)
    vals = mq.max(0)
    matches = mq.argmax(0).astype(np.int64)            # first maximum
    all_matches = matches.copy()
    below = vals < F(low_threshold)
    between = (vals >= F(low_threshold)) & (vals < F(high_threshold))
    matches[below] = BELOW_LOW_THRESHOLD
    matches[between] = BETWEEN_THRESHOLDS
    if allow_low_quality_matches:
        highest = mq.max(1)                            # per ground truth, over the anchors
        upd = (mq == highest[:, None]).any(0)          # every anchor that ties with a row maximum (matcher.py:126-128)
        matches[upd] = all_matches[upd]
        thr = np.maximum(F(0.02), (highest - F(0.05)).astype(F))            # :166-167
        ign = (mq > thr[:, None]).any(0) & (matches == BELOW_LOW_THRESHOLD)  # :168-172
        matches[ign] = BETWEEN_THRESHOLDS
    return matches, vals

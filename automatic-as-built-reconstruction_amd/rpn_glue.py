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
from sparseconvnet import SCN as _SCN
import _nms
from _hip import ptr, stream, check


def grid_anchors(site_coords, base_anchors, voxel_scale, stride):
    """all anchors of a map, flattened [site, yaw] like AnchorGenerator.grid_anchors: [V*A, 7]"""
    c = site_coords[:, 0:3].float() / voxel_scale * torch.as_tensor(stride, dtype=torch.float32,
                                                                    device=site_coords.device).view(1, 3)
    c = torch.cat([c, torch.zeros(c.shape[0], 4, device=c.device)], 1).view(-1, 1, 7)
    return (c + base_anchors.view(1, -1, 7).to(c.device)).reshape(-1, 7)


_anchor_cache = {}


def _device_anchors(base_anchors, A, dev):
    """the maps' base anchors as one [n_maps * A, 7] device tensor; constants of the config, uploaded once per
    (tensor objects, versions, device) -- a pageable host->device copy blocks the host until the stream gets there"""
    key = (tuple((id(b), b._version) for b in base_anchors), str(dev))
    hit = _anchor_cache.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[0], base_anchors)):
        if len(_anchor_cache) > 64:
            _anchor_cache.clear()
        t = torch.cat([b.reshape(A, 7).to(torch.float32).cpu() for b in base_anchors], 0).to(dev).contiguous()
        hit = _anchor_cache[key] = (list(base_anchors), t)
    return hit[1]


def rpn_proposals_single_map(tensor, objectness, box_regression, base_anchors, voxel_scale, stride,
                             pre_nms_top_n=2000, post_nms_top_n=1000, nms_thresh=0.5,
                             nms_aug_thickness=(0.3, 0.3), weights=(1.0,) * 7, bbox_xform_clip=10000.0):
    """tensor: SparseConvNetTensor of the map (sites batch-contiguous); objectness [V*A] logits and
    box_regression [V*A,7] in the flatten order [site, yaw].  Returns a list over examples of
    (boxes [m,7] yx_zb, objectness [m]) after NMS, all on the device."""
    lib = _hip.load()
    g = tensor.metadata.grids[_SCN._key(tensor.spatial_size)]
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


fused_topk = True      # aabr_rpn_topk_maps instead of torch.cat + torch.topk per example (False: the round-4 path, for A/B)
topk_stats = {"fallbacks": 0}
_trace = None     # tools/tools_step_timeline.py: called with a label at the stage's host-side boundaries
debug_nms_inputs = None   # tests: a list here receives (nms_boxes [k,7], scores [k]) of every example's NMS call


def rpn_proposals(maps, objectness, box_regression, base_anchors, strides, voxel_scale, pre_nms_top_n=2000,
                  post_nms_top_n=1000, nms_thresh=0.5, nms_aug_thickness=(0.3, 0.3), weights=(1.0,) * 7,
                  bbox_xform_clip=10000.0, batch_size=None, batched=False, defer=False):
    """Cross-scale proposals, the shape the reference runs in (RPNModule.forward, rpn_sparse3d.py:184-209):
    `cat_scales_obj_reg` regroups the scales example-major and RPNPostProcessor then does, per example, ONE
    sigmoid -> top-k(2000) -> decode -> boxlist_nms_3d(1000) over the anchors of all maps
    (rpn/inference_3d.py:95-149).

    maps: list of SparseConvNetTensor (sites batch-contiguous); objectness[m] [V_m*A] logits and
    box_regression[m] [V_m*A,7] in the flatten order [site, yaw]; base_anchors[m] [A,7]; strides[m] (3).
    Device-resident: neither the anchors nor the regrouped tensors are materialised -- the fused kernel maps a
    selected index of the example's concatenated list back to (map, site, yaw).  The top-k runs on the logits
    (sigmoid is monotone; it is applied to the k selected entries inside the kernel).  One small read-back
    (sites per example and map) sizes the slices.  Returns a list over examples of (boxes [m,7], scores [m])."""
    lib = _hip.load()
    n_maps = len(maps)
    assert n_maps == len(objectness) == len(box_regression) == len(base_anchors) == len(strides)
    grids = [t.metadata.grids[_SCN._key(t.spatial_size)] for t in maps]
    dev = objectness[0].device
    A = int(base_anchors[0].shape[0])
    ba = _device_anchors(base_anchors, A, dev)
    obj = [o.reshape(-1).contiguous().float() for o in objectness]
    reg = [r.reshape(-1, 7).contiguous().float() for r in box_regression]
    # `batch_size` (extension): the number of examples when the caller knows it -- saves the read of the last site's
    # batch index.  `batched` (extension): one top-k over a padded [examples, anchors] matrix and one library call for
    # every example's decode + NMS (aabr_rpn_gather_logits / aabr_rpn_proposals_batch) when every example has at
    # least pre_nms_top_n anchors; same selections unless logits tie exactly at the cut
    if batch_size is not None:
        nb = int(batch_size)
    else:
        nb = max((int(g.coords[-1, 3].item()) + 1 if g.V else 0) for g in grids[:1])
    # sites per (map, example): the per-sample row offsets every grid got with its site-count read
    # (SparseGrid::ctr, Metadata.h:24-33); only a grid built without them costs a read here
    counts = [g.sample_counts(nb) for g in grids]
    if any(c is None for c in counts):
        counts = torch.stack([torch.bincount(g.coords[:, 3].long(), minlength=nb)[:nb] if g.V else
                              torch.zeros(nb, dtype=torch.int64, device=dev) for g in grids]).tolist()
    if batched and 1 <= nb <= 16 and n_maps <= 8:
        segs, sites, site0 = [], [], [0] * n_maps
        for bi in range(nb):
            seg = [0]
            for m in range(n_maps):
                seg.append(seg[-1] + counts[m][bi] * A)
            segs.append(seg)
            sites.append(list(site0))
            for m in range(n_maps):
                site0[m] += counts[m][bi]
        if min(s[-1] for s in segs) >= pre_nms_top_n:
            k = int(pre_nms_top_n)
            lmax = max(s[-1] for s in segs)
            seg_h = _hip.i32xn([v for s_ in segs for v in s_])
            site_h = _hip.i32xn([v for s_ in sites for v in s_])
            padded = torch.empty((nb, lmax), dtype=torch.float32, device=dev)
            check(lib.aabr_rpn_gather_logits(n_maps, _hip.ptrs(obj), nb, seg_h, site_h, A, lmax, ptr(padded), stream()))
            _, sel = padded.topk(k, dim=1, sorted=True)
            boxes = torch.empty((nb, k, 7), dtype=torch.float32, device=dev)
            nms_boxes = torch.empty((nb, k, 7), dtype=torch.float32, device=dev)
            scores = torch.empty((nb, k), dtype=torch.float32, device=dev)
            cb = (k + 63) // 64
            mask = torch.empty(nb * k * cb, dtype=torch.int64, device=dev)
            keep = torch.empty((nb, k), dtype=torch.int64, device=dev)
            meta = torch.empty((nb, _hip.META_WORDS), dtype=torch.int32, device=dev)
            check(lib.aabr_rpn_proposals_batch(
                n_maps, _hip.ptrs([g.coords for g in grids]), _hip.ptrs(obj), _hip.ptrs(reg), nb, seg_h, site_h,
                _hip.f32xn([v for st in strides for v in st]), ptr(ba), A, float(voxel_scale), _hip.f32xn(weights),
                float(bbox_xform_clip), float(nms_aug_thickness[0]), float(nms_aug_thickness[1]), ptr(sel), k,
                ptr(boxes), ptr(nms_boxes), ptr(scores), float(nms_thresh), int(_nms.REFERENCE_DEBUG_ONLY_XY),
                int(post_nms_top_n), ptr(mask), ptr(keep), ptr(meta), stream()))
            kept = _hip.read_back(meta[:, 0])       # the one read of the stage
            out = []
            for bi in range(nb):
                kk = keep[bi, :kept[bi]]
                out.append((boxes[bi][kk], scores[bi][kk]))
            return out
    coords_p, obj_p, reg_p = _hip.ptrs([g.coords for g in grids]), _hip.ptrs(obj), _hip.ptrs(reg)
    strides_h = _hip.f32xn([v for st in strides for v in st])
    weights_h = _hip.f32xn(weights)
    # fused cross-scale top-k (csrc/iou_nms.hip aabr_rpn_topk_maps): every example's selection in four launches, nothing
    # concatenated; torch.topk per example (a chain of ~9 rocprim launches each, on a concatenated copy) stays as the
    # fallback for shapes the kernel does not take and for the (reported) case of > 4096 exactly tied logits at the cut
    sel_all = topk_info = None
    if fused_topk and 1 <= nb <= 16 and n_maps <= 8 and pre_nms_top_n <= 2048:
        segs, sites, s0 = [], [], [0] * n_maps
        for bi in range(nb):
            seg = [0]
            for m in range(n_maps):
                seg.append(seg[-1] + counts[m][bi] * A)
            segs += seg
            sites += s0
            s0 = [s0[m] + counts[m][bi] for m in range(n_maps)]
        ks = [min(pre_nms_top_n, segs[bi * (n_maps + 1) + n_maps]) for bi in range(nb)]
        kmax = max(max(ks), 1)
        sel_all = torch.empty((nb, kmax), dtype=torch.int64, device=dev)
        topk_info = torch.empty((nb, 2), dtype=torch.int32, device=dev)
        tscr = _hip.workspace("rpn_topk", int(lib.aabr_rpn_topk_scratch_words(nb)) + 2, torch.int32, dev)
        tso = (-tscr.data_ptr() // 4) % 2
        check(lib.aabr_rpn_topk_maps(n_maps, obj_p, nb, _hip.i32xn(segs), _hip.i32xn(sites), A, _hip.i32xn(ks), ptr(sel_all),
                                     kmax, ptr(topk_info), tscr.data_ptr() + 4 * tso, stream()))
    site0 = [0] * n_maps
    out, pending = [], []
    for bi in range(nb):
        seg = [0]
        for m in range(n_maps):
            seg.append(seg[-1] + counts[m][bi] * A)
        n_anchor = seg[-1]
        if n_anchor == 0:
            out.append((torch.zeros(0, 7, device=dev), torch.zeros(0, device=dev)))
            continue
        k = min(pre_nms_top_n, n_anchor)
        if sel_all is not None:
            sel = sel_all[bi, :k]
        else:
            logit_b = torch.cat([obj[m][site0[m] * A:(site0[m] + counts[m][bi]) * A] for m in range(n_maps)])
            _, sel = logit_b.topk(k, dim=0, sorted=True)
        boxes = torch.empty((k, 7), dtype=torch.float32, device=dev)
        nms_boxes = torch.empty((k, 7), dtype=torch.float32, device=dev)
        scores = torch.empty(k, dtype=torch.float32, device=dev)
        check(lib.aabr_rpn_decode_maps(n_maps, coords_p, obj_p, reg_p, _hip.i32xn(seg), _hip.i32xn(site0), strides_h,
                                       ptr(ba), A, float(voxel_scale), weights_h, float(bbox_xform_clip),
                                       float(nms_aug_thickness[0]), float(nms_aug_thickness[1]), ptr(sel), k,
                                       ptr(boxes), ptr(nms_boxes), ptr(scores), stream()))
        if debug_nms_inputs is not None:
            debug_nms_inputs.append((nms_boxes, scores))
        # every example's launches go out first; the numbers kept are read once, after the last one
        keep, meta = _nms.rotate_nms_sorted(nms_boxes, nms_thresh, post_nms_top_n, _nms.REFERENCE_DEBUG_ONLY_XY,
                                            lazy=True)
        pending.append((len(out), boxes, scores, keep, meta, bi, list(seg), list(site0)))
        out.append(None)
        for m in range(n_maps):
            site0[m] += counts[m][bi]
    def finish():
        if pending:
            if _trace is not None:
                _trace("proposal launches enqueued")
            words = [p[4][0:1] for p in pending]
            if topk_info is not None:
                words.append(topk_info[:, 1])
            vals = _hip.read_back(torch.cat(words))                             # the one read of the stage
            kept, over = vals[:len(pending)], vals[len(pending):]
            if _trace is not None:
                _trace("proposal counts read")
            for (i, boxes, scores, keep, meta, bi, seg, st0), nk in zip(pending, kept):
                if over and over[bi]:
                    # > 4096 exactly tied logits at the cut: this example again, selected by a full torch.topk
                    topk_stats["fallbacks"] += 1
                    logit_b = torch.cat([obj[m][st0[m] * A:st0[m] * A + (seg[m + 1] - seg[m])] for m in range(n_maps)])
                    k_ = boxes.shape[0]
                    _, sel = logit_b.topk(k_, dim=0, sorted=True)
                    nms_boxes = torch.empty((k_, 7), dtype=torch.float32, device=dev)
                    check(lib.aabr_rpn_decode_maps(n_maps, coords_p, obj_p, reg_p, _hip.i32xn(seg), _hip.i32xn(st0), strides_h,
                                                   ptr(ba), A, float(voxel_scale), weights_h, float(bbox_xform_clip),
                                                   float(nms_aug_thickness[0]), float(nms_aug_thickness[1]), ptr(sel), k_,
                                                   ptr(boxes), ptr(nms_boxes), ptr(scores), stream()))
                    kp = _nms.rotate_nms_sorted(nms_boxes, nms_thresh, post_nms_top_n, _nms.REFERENCE_DEBUG_ONLY_XY)
                    out[i] = (boxes[kp], scores[kp])
                    continue
                k = keep[:nk]
                out[i] = (boxes[k], scores[k])
        return out

    # `defer` (extension): every launch of the stage is out; the caller gets the function that does the one read and
    # slices the lists, to be called (on the same stream) when it wants the result -- e.g. after it has enqueued
    # other work that does not depend on the proposals
    return finish if defer else finish()


def rpn_label_matches(maps, base_anchors, strides, voxel_scale, targets, aug_thickness, criterion=6,
                      fg_iou=0.55, bg_iou=0.2, batch_size=None, return_matrix=False, yaw_threshold=0.7,
                      allow_low_quality_matches=True, regression_targets=False, weights=(1.0,) * 7):
    """The label-generation half of the RPN's training step on the device: per example the IoU of its ground-truth
    boxes against the anchors of ALL maps, `boxlist_iou_3d(target, anchor, aug_thickness, criterion,
    flag='rpn_label_generation')` (RPNLossComputation.match_targets_to_anchors, modeling/rpn/loss_3d.py:91-100;
    criterion = cfg.MODEL.IOU_CRITERIA = 6, config/defaults.py:44), followed by `Matcher.__call__` as
    make_rpn_loss_evaluator builds it (loss_3d.py:338-344, modeling/matcher.py:50-196): entries whose |yaw
    difference| is not below `yaw_threshold` (cfg.MODEL.RPN.YAW_THRESHOLD = 0.7, defaults.py:153; > 1.58 = no mask)
    are zeroed, best ground truth per anchor, BELOW_LOW_THRESHOLD (-1) / BETWEEN_THRESHOLDS (-2) by the two IoU
    thresholds (defaults.py:147,151), then -- `allow_low_quality_matches`, True for the RPN -- set_low_quality_matches_
    and its ignore-nearby pass.  The `cendis` argument the reference also passes is dead there (`if cendis is None or
    True`, matcher.py:130).  The sampler and the losses are plain torch in the reference and are not part of this path.
    `matched_vals` / the returned matrix: the masked maximum / the UNMASKED `boxlist_iou_3d` matrix.

    maps / base_anchors / strides as in `rpn_proposals`; targets[b] = [G_b, 7] yx_zb boxes of example b (device).
    ONE library call for the whole batch (`aabr_rpn_label_generation`): anchors are generated inside the kernel from
    the maps' site lists (AnchorGenerator.grid_anchors, anchor_generator_sparse3d.py:88-104, example-major like
    `cat_scales_anchor`), the per-example row ranges come from the grids' per-sample offsets (no host read), the
    [G_b, N_b] matrices are written only with `return_matrix`.
    `regression_targets` True: the same launch also writes what RPNLossComputation.prepare_targets computes next
    (loss_3d.py:186-196) -- `box_coder.encode(target[matched_idxs.clamp(min=0)], anchor)` for every anchor, BoxCoder3D's
    centroid form with `weights` (box_coder_3d.py:46-51; an example without ground truth encodes its anchors against
    themselves, loss_3d.py:91-94) -- and every tuple gets it as a fourth entry, fp32 [N_b, 7].
    Returns a list over examples of (matched_idxs int64 [N_b], matched_vals fp32 [N_b], iou [G_b, N_b] or None
    [, regression targets])."""
    from utils3d import rotate_nms_3d_torch as R
    lib = _hip.load()
    n_maps = len(maps)
    grids = [t.metadata.grids[_SCN._key(t.spatial_size)] for t in maps]
    dev = grids[0].coords.device
    A = int(base_anchors[0].shape[0])
    ba = _device_anchors(base_anchors, A, dev)
    nb = int(batch_size) if batch_size is not None else len(targets)
    assert aug_thickness["anchor_Y"] == 0 and aug_thickness["target_Y"] >= 0.3     # rotate_nms_3d_torch.py:34-36
    counts = [g.sample_counts(nb) for g in grids]
    if any(c is None for c in counts):
        counts = torch.stack([torch.bincount(g.coords[:, 3].long(), minlength=nb)[:nb] if g.V else
                              torch.zeros(nb, dtype=torch.int64, device=dev) for g in grids]).tolist()
    out = []
    for b0 in range(0, nb, 16):            # the library takes up to 16 examples per call
        b1 = min(nb, b0 + 16)
        seg, site, site0 = [], [], [sum(counts[m][:b0]) for m in range(n_maps)]
        n_anch, tg = [], []
        for bi in range(b0, b1):
            s_ = [0]
            for m in range(n_maps):
                s_.append(s_[-1] + counts[m][bi] * A)
            seg += s_
            site += list(site0)
            for m in range(n_maps):
                site0[m] += counts[m][bi]
            n_anch.append(s_[-1])
            t = targets[bi].to(device=dev, dtype=torch.float32).contiguous()
            tg.append(t)
        total = sum(n_anch)
        midx = torch.empty(total, dtype=torch.int64, device=dev)
        mval = torch.empty(total, dtype=torch.float32, device=dev)
        mat = torch.empty(sum(n * int(t.shape[0]) for n, t in zip(n_anch, tg)), dtype=torch.float32,
                          device=dev) if return_matrix else None
        aug = (aug_thickness["target_Y"], aug_thickness["target_Z"], aug_thickness["anchor_Y"],
               aug_thickness["anchor_Z"])
        n_gt = sum(int(t.shape[0]) for t in tg)
        rowmax = torch.empty(max(n_gt, 1), dtype=torch.int32, device=dev) if allow_low_quality_matches else None
        regt = torch.empty((total, 7), dtype=torch.float32, device=dev) if regression_targets else None
        check(lib.aabr_rpn_label_generation_targets(
            n_maps, _hip.ptrs([g.coords for g in grids]), b1 - b0, _hip.i32xn(seg), _hip.i32xn(site),
            _hip.f32xn([v for st in strides for v in st]), ptr(ba), A, float(voxel_scale), _hip.ptrs(tg),
            _hip.i32xn([int(t.shape[0]) for t in tg]), _hip.f32x4(aug), int(criterion), int(bool(R.DEBUG)),
            float(fg_iou), float(bg_iou), float(yaw_threshold), int(bool(allow_low_quality_matches)), ptr(midx),
            ptr(mval), ptr(mat), ptr(rowmax), _hip.f32xn(weights), ptr(regt), stream()))
        o = mo = 0
        for n, t in zip(n_anch, tg):
            G = int(t.shape[0])
            row = (midx[o:o + n], mval[o:o + n], mat[mo:mo + G * n].view(G, n) if return_matrix else None)
            out.append(row + (regt[o:o + n],) if regression_targets else row)
            o += n
            mo += G * n
    return out


_stride_cache = {}


def _stride_t(stride, dev):
    key = (tuple(float(v) for v in stride), str(dev))
    t = _stride_cache.get(key)
    if t is None:
        t = _stride_cache[key] = torch.tensor(key[0], dtype=torch.float32).to(dev).view(1, 3)
    return t


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

"""ctypes binding of oracle/liboracle.so -- tests / smoke / cpu_baseline ONLY."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_lib = None

i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "liboracle.so")])
        _lib = C.CDLL(so)
        L = _lib
        L.oracle_input_layer_sites.restype = C.c_int64
        L.oracle_input_layer_sites.argtypes = [i64p, C.c_int64, C.c_int, i32p, i64p, i32p, C.POINTER(C.c_int32)]
        L.oracle_input_layer_rules.argtypes = [i32p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, i32p]
        L.oracle_input_layer_fwd.argtypes = [f32p, f32p, C.c_int64, C.c_int32, C.c_int32, i32p, C.c_int]
        L.oracle_input_layer_bwd.argtypes = [f32p, f32p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, i32p, C.c_int]
        L.oracle_submanifold_rules.restype = C.c_int64
        L.oracle_submanifold_rules.argtypes = [i64p, C.c_int64, i64p, i32p, i64p]
        L.oracle_convolution_rules.restype = C.c_int64
        L.oracle_convolution_rules.argtypes = [i64p, C.c_int64, i64p, i64p, i64p, C.c_int64, i32p, i64p, i64p]
        L.oracle_conv_fwd.restype = C.c_double
        L.oracle_conv_fwd.argtypes = [f32p, C.c_int32, f32p, C.c_int32, C.c_int64, f32p, C.c_void_p, i32p, i64p,
                                      C.c_int64, C.c_int64, C.c_int]
        L.oracle_conv_bwd.argtypes = [f32p, f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, f32p, f32p,
                                      C.c_void_p, i32p, i64p, C.c_int64, C.c_int64, C.c_int]
        L.oracle_bn_fwd.argtypes = [f32p, f32p, C.c_int32, C.c_int64, f32p, f32p, f32p, f32p, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_float, C.c_int, C.c_float]
        L.oracle_bn_bwd.argtypes = [f32p, f32p, f32p, f32p, C.c_int32, C.c_int64, f32p, f32p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_rotate_iou_eval.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int, f32p]
        L.oracle_boxes_iou_3d.argtypes = [f32p, C.c_int64, f32p, C.c_int64, f32p, C.c_int, C.c_int, f32p]
        L.oracle_nms_from_matrix.restype = C.c_int64
        L.oracle_nms_from_matrix.argtypes = [f32p, C.c_int64, i32p, C.c_float, i64p]
        L.oracle_nms_prefilter_decide.restype = C.c_int64
        L.oracle_nms_prefilter_decide.argtypes = [f32p, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"),
                                                  C.c_int64, i32p, C.c_float, i64p]
        L.oracle_nms_axis_aligned.restype = C.c_int64
        L.oracle_nms_axis_aligned.argtypes = [f32p, f32p, C.c_int64, C.c_float, i64p]
        L.oracle_sparse_to_dense_fwd.argtypes = [i64p, C.c_int64, f32p, C.c_int, i64p, C.c_int64, f32p]
        L.oracle_sparse_to_dense_bwd.argtypes = [i64p, C.c_int64, f32p, C.c_int, i64p, f32p]
        L.oracle_roi_align_rot3d.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int]
        L.oracle_clip_iou_matrix.argtypes = [f32p, C.c_int64, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_region_points.restype = C.c_int64
        L.oracle_region_points.argtypes = [i64p, i64p, i64p]
        L.oracle_region_offset.restype = C.c_int32
        L.oracle_region_offset.argtypes = [i64p, i64p, i64p]
        for f in ("oracle_input_region",):
            getattr(L, f).argtypes = [i64p, i64p, i64p, i64p, i64p]
        L.oracle_output_region.argtypes = [i64p, i64p, i64p, i64p, i64p, i64p]
        L.oracle_submanifold_region.argtypes = [i64p, i64p, i64p, i64p]
    return _lib


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return lib().oracle_num_threads()


def set_threads(n):
    lib().oracle_set_threads(int(n))
    return num_threads()


# ---------------------------------------------------------------- input layer
def input_layer(coords, feats, mode=4):
    """-> dict(out[V,C], coords[V,4], point_voxel[n], counts[V], rules[V,1+ma], max_active)"""
    coords = np.ascontiguousarray(coords, np.int64)
    n, ncols = coords.shape
    pv = np.zeros(n, np.int32)
    oc = np.zeros((max(n, 1), 4), np.int64)
    cnt = np.zeros(max(n, 1), np.int32)
    ma = C.c_int32(0)
    V = lib().oracle_input_layer_sites(coords, n, ncols, pv, oc, cnt, C.byref(ma))
    if V < 0:
        raise ValueError("oracle_input_layer_sites failed: %d" % V)
    ma = ma.value
    w = (ma if mode in (3, 4) else 1) + 1
    rules = np.zeros((V, w), np.int32)
    lib().oracle_input_layer_rules(pv, n, V, mode, ma, rules)
    out = None
    if feats is not None:
        feats = np.ascontiguousarray(feats, np.float32)
        out = np.zeros((V, feats.shape[1]), np.float32)
        lib().oracle_input_layer_fwd(feats, out, V, w - 1, feats.shape[1], rules, int(mode == 4))
    return dict(out=out, coords=oc[:V].copy(), point_voxel=pv, counts=cnt[:V].copy(), rules=rules,
                max_active=w - 1, V=V, mode=mode, n=n)


def input_layer_bwd(il, d_out):
    d_out = np.ascontiguousarray(d_out, np.float32)
    d_in = np.zeros((il["n"], d_out.shape[1]), np.float32)
    lib().oracle_input_layer_bwd(d_in, d_out, il["n"], il["V"], il["max_active"], d_out.shape[1], il["rules"],
                                 int(il["mode"] == 4))
    return d_in


# ---------------------------------------------------------------- rule books
class Rules:
    """rules[vol, cap, 2] int32 + counts[vol]; .pairs(k) -> [n_k, 2]"""

    def __init__(self, rules, counts, cap):
        self.rules, self.counts, self.cap = rules, counts, cap
        self.vol = counts.shape[0]

    def pairs(self, k):
        return self.rules[k, : self.counts[k]]

    @property
    def total(self):
        return int(self.counts.sum())


def submanifold_rules(site_coords, size):
    sc = np.ascontiguousarray(site_coords, np.int64)
    V = sc.shape[0]
    size = np.asarray(size, np.int64)
    vol = int(size.prod())
    rules = np.zeros((vol, max(V, 1), 2), np.int32)
    counts = np.zeros(vol, np.int64)
    tot = lib().oracle_submanifold_rules(sc, V, size, rules, counts)
    assert tot == counts.sum()
    return Rules(rules, counts, max(V, 1))


def convolution_rules(in_coords, size, stride, out_spatial):
    ic = np.ascontiguousarray(in_coords, np.int64)
    V = ic.shape[0]
    size = np.asarray(size, np.int64)
    stride = np.asarray(stride, np.int64)
    out_spatial = np.asarray(out_spatial, np.int64)
    vol = int(size.prod())
    maxout = int(np.prod((size + stride - 1) // stride))
    cap = max(V, 1)
    rules = np.zeros((vol, cap, 2), np.int32)
    counts = np.zeros(vol, np.int64)
    oc = np.zeros((max(V * maxout, 1), 4), np.int64)
    Vo = lib().oracle_convolution_rules(ic, V, size, stride, out_spatial, cap, rules, counts, oc)
    if Vo < 0:
        raise ValueError("oracle_convolution_rules failed: %d" % Vo)
    return Rules(rules, counts, cap), oc[:Vo].copy()


def conv_fwd(inp, W, rb, n_out_rows, bias=None, in_col=0):
    inp = np.ascontiguousarray(inp, np.float32)
    W = np.ascontiguousarray(W, np.float32).reshape(rb.vol, inp.shape[1], -1)
    nOut = W.shape[2]
    out = np.zeros((n_out_rows, nOut), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    macs = lib().oracle_conv_fwd(inp, inp.shape[1], out, nOut, n_out_rows, W, _opt(b), rb.rules, rb.counts, rb.vol,
                                 rb.cap, in_col)
    return out, macs


def conv_bwd(inp, d_out, W, rb, in_col=0, want_bias=False):
    inp = np.ascontiguousarray(inp, np.float32)
    d_out = np.ascontiguousarray(d_out, np.float32)
    W = np.ascontiguousarray(W, np.float32).reshape(rb.vol, inp.shape[1], -1)
    d_in = np.zeros_like(inp)
    dW = np.zeros_like(W)
    db = np.zeros(W.shape[2], np.float32) if want_bias else None
    lib().oracle_conv_bwd(inp, d_in, inp.shape[0], inp.shape[1], d_out, d_out.shape[0], d_out.shape[1], W, dW,
                          _opt(db), rb.rules, rb.counts, rb.vol, rb.cap, in_col)
    return d_in, dW, db


# ---------------------------------------------------------------- batch norm
def bn_fwd(x, weight, bias, running_mean, running_var, eps=1e-4, momentum=0.9, train=True, leakiness=0.0):
    x = np.ascontiguousarray(x, np.float32)
    C_ = x.shape[1]
    out = np.zeros_like(x)
    sm, si = np.zeros(C_, np.float32), np.zeros(C_, np.float32)
    rm, rv = running_mean.astype(np.float32).copy(), running_var.astype(np.float32).copy()
    w = None if weight is None else np.ascontiguousarray(weight, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().oracle_bn_fwd(x, out, C_, x.shape[0], sm, si, rm, rv, _opt(w), _opt(b), eps, momentum, int(train),
                        leakiness)
    return out, sm, si, rm, rv


def bn_bwd(x, out, d_out, save_mean, save_invstd, weight, leakiness=0.0):
    x = np.ascontiguousarray(x, np.float32)
    d_out = np.ascontiguousarray(d_out, np.float32).copy()
    C_ = x.shape[1]
    d_in = np.zeros_like(x)
    dw, db = np.zeros(C_, np.float32), np.zeros(C_, np.float32)
    w = None if weight is None else np.ascontiguousarray(weight, np.float32)
    lib().oracle_bn_bwd(x, d_in, np.ascontiguousarray(out, np.float32), d_out, C_, x.shape[0],
                        np.ascontiguousarray(save_mean, np.float32), np.ascontiguousarray(save_invstd, np.float32),
                        _opt(w), _opt(dw), _opt(db), leakiness)
    return d_in, dw, db, d_out


# ---------------------------------------------------------------- IoU / NMS
def rotate_iou_eval(boxes, query, criterion=-1):
    boxes = np.ascontiguousarray(boxes, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    iou = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    if iou.size:
        lib().oracle_rotate_iou_eval(boxes, boxes.shape[0], query, query.shape[0], criterion, iou)
    return iou


def boxes_iou_3d(targets, anchors, aug=(0, 0, 0, 0), criterion=-1, only_xy=True):
    t = np.ascontiguousarray(targets, np.float32)
    a = np.ascontiguousarray(anchors, np.float32)
    iou = np.zeros((t.shape[0], a.shape[0]), np.float32)
    if iou.size:
        lib().oracle_boxes_iou_3d(t, t.shape[0], a, a.shape[0], np.asarray(aug, np.float32), criterion, int(only_xy),
                                  iou)
    return iou


def clip_iou_matrix(boxes7):
    """exact fp64 2-D IoU of rotated rectangles by Sutherland-Hodgman clipping (oracle/clip_oracle.c) -- the value
    a boost::geometry polygon IoU (spconv 1.x's suppression loop) decides on; independent of the A15 restatement"""
    b = np.ascontiguousarray(boxes7, np.float32)
    out = np.zeros((b.shape[0], b.shape[0]), np.float64)
    if out.size:
        lib().oracle_clip_iou_matrix(b, b.shape[0], out)
    return out


def nms_from_matrix(iou, order, thresh):
    n = iou.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = lib().oracle_nms_from_matrix(np.ascontiguousarray(iou, np.float32), n,
                                      np.ascontiguousarray(order, np.int32), thresh, keep)
    return keep[:nk].copy()


def nms_prefilter_decide(pre, dec, order, thresh):
    """greedy loop of spconv 1.x's rotate_non_max_suppression_cpu: `pre` > 0 pre-filter (the A15 matrix the
    reference passes), decision `dec` >= thresh (exact polygon IoU)"""
    n = pre.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = lib().oracle_nms_prefilter_decide(np.ascontiguousarray(pre, np.float32),
                                           np.ascontiguousarray(dec, np.float64), n,
                                           np.ascontiguousarray(order, np.int32), thresh, keep)
    return keep[:nk].copy()


def rotate_nms_3d(boxes7, scores, pre_max_size, post_max_size, thresh, only_xy=True, decision="clip"):
    """box_torch_ops.py:557-582 + nms_cpu.py:32-44: the oracle's A15 matrix as the `> 0` pre-filter and -- the
    stated rule for the un-vendored spconv loop, DESIGN.md section 4 -- the exact polygon IoU (clip_oracle.c) for
    the `>= thresh` decision.  decision="matrix": the round-1..3 rule (decision on the A15 value itself), kept to
    count how often the two differ."""
    scores = np.asarray(scores, np.float32)
    n = scores.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    if pre_max_size is not None:
        k = min(n, pre_max_size)
        idx = np.argsort(-scores, kind="stable")[:k]
    else:
        idx = np.arange(n)
    b = np.asarray(boxes7, np.float32)[idx]
    s = scores[idx]
    iou = boxes_iou_3d(b, b, (0, 0, 0, 0), -1, only_xy)
    order = np.argsort(-s, kind="stable").astype(np.int32)
    if decision == "clip":
        keep = nms_prefilter_decide(iou, clip_iou_matrix(b), order, thresh)[:post_max_size]
    else:
        keep = nms_from_matrix(iou, order, thresh)[:post_max_size]
    return idx[keep]


def nms_axis_aligned(dets, scores, thresh):
    dets = np.ascontiguousarray(dets, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    keep = np.zeros(max(len(scores), 1), np.int64)
    nk = lib().oracle_nms_axis_aligned(dets, scores, len(scores), thresh, keep)
    return keep[:nk].copy()


# ---------------------------------------------------------------- SparseToDense / ROI align
def sparse_to_dense(site_coords, feats, spatial, batch):
    sc = np.ascontiguousarray(site_coords, np.int64)
    f = np.ascontiguousarray(feats, np.float32)
    sp = np.asarray(spatial, np.int64)
    out = np.zeros((batch, f.shape[1]) + tuple(int(v) for v in sp), np.float32)
    lib().oracle_sparse_to_dense_fwd(sc, sc.shape[0], f, f.shape[1], sp, batch, out)
    return out


def sparse_to_dense_bwd(site_coords, d_out, planes, spatial):
    sc = np.ascontiguousarray(site_coords, np.int64)
    d_in = np.zeros((sc.shape[0], planes), np.float32)
    lib().oracle_sparse_to_dense_bwd(sc, sc.shape[0], d_in, planes, np.asarray(spatial, np.int64),
                                     np.ascontiguousarray(d_out, np.float32))
    return d_in


def roi_align_rot3d_fwd(inp, rois, scale, out_size, sampling):
    inp = np.ascontiguousarray(inp, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, Cc, H, W, Z = inp.shape
    out = np.zeros((rois.shape[0], Cc) + tuple(out_size), np.float32)
    lib().oracle_roi_align_rot3d(_opt(inp), rois, rois.shape[0], scale, Cc, H, W, Z, out_size[0], out_size[1],
                                 out_size[2], sampling, _opt(out), None, None, 0)
    return out


def roi_align_rot3d_bwd(grad, rois, scale, out_size, shape, sampling):
    grad = np.ascontiguousarray(grad, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, Cc, H, W, Z = shape
    gin = np.zeros(shape, np.float32)
    lib().oracle_roi_align_rot3d(None, rois, rois.shape[0], scale, Cc, H, W, Z, out_size[0], out_size[1],
                                 out_size[2], sampling, None, _opt(grad), _opt(gin), 1)
    return gin

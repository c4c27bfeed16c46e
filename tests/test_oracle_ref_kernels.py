"""Pins the NUMERICAL half of oracle/scn_oracle.c to the reference's OWN CPU kernels
(SCN/CPU/BatchNormalization.cpp, IOLayers.cpp, Convolution.cpp gather/scatter + at::matmul,
SparseToDense.cpp), compiled from /root/reference where they lie into oracle/_ref/libref_kernels.so
(recipe: oracle/Makefile, harness: oracle/ref_kernels_harness.cpp).  The .so travels to the GPU box
with the repo snapshot; /root/reference itself is never read at test time.

Bars: BatchNorm fwd/bwd and InputLayer fwd/bwd BIT-EQUAL (same sequential fp32 arithmetic);
conv / deconv fwd + bwd within 1e-5 relative (the reference contracts with at::matmul = MKL sgemm,
whose summation order is not specified; the oracle accumulates in double); SparseToDense exact."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import synth_scenes as S

SO = os.path.join(O.ORACLE_DIR, "_ref", "libref_kernels.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref not built (needs /root/reference)")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
vp = C.c_void_p


def _opt(a):
    return None if a is None else a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(SO)
    L.ref_bn_fwd.argtypes = [f32p, f32p, C.c_int, C.c_long, f32p, f32p, f32p, f32p, vp, vp, C.c_float, C.c_float,
                             C.c_int, C.c_float]
    L.ref_bn_bwd.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_long, f32p, f32p, f32p, f32p, vp, vp, vp, vp,
                             C.c_float]
    L.ref_input_layer_fwd.argtypes = [f32p, f32p, C.c_long, C.c_int, C.c_int, i32p, C.c_int]
    L.ref_input_layer_bwd.argtypes = [f32p, f32p, C.c_long, C.c_int, C.c_int, i32p, C.c_int]
    L.ref_rule_conv_fwd.restype = C.c_double
    L.ref_rule_conv_fwd.argtypes = [f32p, C.c_long, C.c_int, f32p, C.c_long, C.c_int, f32p, i32p, i64p, C.c_long,
                                    C.c_long, C.c_int]
    L.ref_rule_conv_bwd.argtypes = [f32p, f32p, C.c_long, C.c_int, f32p, C.c_long, C.c_int, f32p, f32p, i32p, i64p,
                                    C.c_long, C.c_long, C.c_int]
    L.ref_sparse_to_dense_fwd.argtypes = [f32p, vp, C.c_int, C.c_long, i32p, C.c_int]
    L.ref_sparse_to_dense_bwd.argtypes = [f32p, vp, C.c_int, C.c_long, i32p, C.c_int]
    return L


# ------------------------------------------------------------------ BatchNorm: bit-equal
@pytest.mark.parametrize("rows,planes", [(1, 4), (2, 32), (777, 32), (5000, 128), (66094, 32)])
@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("leak", [0.0, 0.333])
def test_bn_forward_backward_bit_equal(ref, rows, planes, affine, leak):
    rng = np.random.default_rng(rows * 131 + planes)
    x = (rng.standard_normal((rows, planes)) * 2 + 0.3).astype(np.float32)
    w = rng.uniform(0.5, 1.5, planes).astype(np.float32) if affine else None
    b = rng.standard_normal(planes).astype(np.float32) if affine else None
    rm0 = rng.standard_normal(planes).astype(np.float32)
    rv0 = rng.uniform(0.5, 2, planes).astype(np.float32)
    for train in ([True, False] if rows > 1 else [False]):
        o_out, o_sm, o_si, o_rm, o_rv = O.bn_fwd(x, w, b, rm0, rv0, 1e-4, 0.95, train, leak)
        r_out = np.zeros_like(x)
        r_sm, r_si = np.zeros(planes, np.float32), np.zeros(planes, np.float32)
        r_rm, r_rv = rm0.copy(), rv0.copy()
        ref.ref_bn_fwd(x.copy(), r_out, planes, rows, r_sm, r_si, r_rm, r_rv, _opt(w), _opt(b), 1e-4, 0.95,
                       int(train), leak)
        for a, bb, name in ((o_out, r_out, "out"), (o_sm, r_sm, "saveMean"), (o_si, r_si, "saveInvStd"),
                            (o_rm, r_rm, "runningMean"), (o_rv, r_rv, "runningVar")):
            assert np.array_equal(a.view(np.uint32), bb.view(np.uint32)), (name, train)
        # backward on the reference's own forward state
        d_out = rng.standard_normal((rows, planes)).astype(np.float32)
        o_din, o_dw, o_db, o_dmask = O.bn_bwd(x, o_out, d_out, o_sm, o_si, w, leak)
        r_din = np.zeros_like(x)
        r_dout = d_out.copy()
        r_dw, r_db = np.zeros(planes, np.float32), np.zeros(planes, np.float32)
        ref.ref_bn_bwd(x.copy(), r_din, r_out, r_dout, planes, rows, r_sm, r_si, r_rm, r_rv, _opt(w), _opt(b),
                       _opt(r_dw), _opt(r_db), leak)
        assert np.array_equal(o_din.view(np.uint32), r_din.view(np.uint32)), train
        assert np.array_equal(o_dmask.view(np.uint32), r_dout.view(np.uint32)), train   # in-place mask
        assert np.array_equal(o_dw.view(np.uint32), r_dw.view(np.uint32)), train
        assert np.array_equal(o_db.view(np.uint32), r_db.view(np.uint32)), train


# ------------------------------------------------------------------ InputLayer: bit-equal
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("npts,scale", [(300, 4), (6000, 20), (79998, 20)])
def test_input_layer_forward_backward_bit_equal(ref, mode, npts, scale):
    locs, feats = S.make_batch(2 if npts < 50000 else 1, npts, 3, scale)
    il = O.input_layer(locs, feats, mode)
    V, ma, Cc = il["V"], il["max_active"], feats.shape[1]
    r_out = np.zeros((V, Cc), np.float32)
    ref.ref_input_layer_fwd(np.ascontiguousarray(feats, np.float32), r_out, V, ma, Cc, il["rules"], int(mode == 4))
    assert np.array_equal(il["out"].view(np.uint32), r_out.view(np.uint32))
    rng = np.random.default_rng(5)
    d_out = rng.standard_normal((V, Cc)).astype(np.float32)
    o_din = O.input_layer_bwd(il, d_out)
    r_din = np.zeros((locs.shape[0], Cc), np.float32)
    ref.ref_input_layer_bwd(r_din, d_out, V, ma, Cc, il["rules"], int(mode == 4))
    assert np.array_equal(o_din.view(np.uint32), r_din.view(np.uint32))


def test_input_layer_empty(ref):
    out = np.zeros((0, 9), np.float32)
    ref.ref_input_layer_fwd(np.zeros((0, 9), np.float32), out, 0, 1, 9, np.zeros((0, 2), np.int32), 1)


# ------------------------------------------------------------------ conv / deconv: 1e-5
def _scene(npts, scale, seed=0, bs=2):
    locs, feats = S.make_batch(bs, npts, seed, scale)
    il = O.input_layer(locs, feats, 4)
    return il


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("nin,nout,filt", [(9, 32, 3), (32, 32, 3), (32, 64, 3), (64, 128, 1), (16, 16, (3, 1, 5))])
def test_submanifold_conv_matches_reference_kernels(ref, nin, nout, filt):
    il = _scene(4000, 20)
    V = il["V"]
    size = [filt] * 3 if isinstance(filt, int) else list(filt)
    rb = O.submanifold_rules(il["coords"], size)
    rng = np.random.default_rng(nin * 7 + nout)
    x = rng.standard_normal((V, nin)).astype(np.float32)
    W = (rng.standard_normal((rb.vol, nin, nout)) * np.sqrt(2.0 / (nin * rb.vol))).astype(np.float32)
    o_out, o_macs = O.conv_fwd(x, W, rb, V)
    r_out = np.zeros((V, nout), np.float32)
    r_macs = ref.ref_rule_conv_fwd(x, V, nin, r_out, V, nout, W, rb.rules, rb.counts, rb.vol, rb.cap, 0)
    assert o_macs == r_macs == float(rb.total) * nin * nout
    assert _rel(o_out, r_out) < 1e-5
    d_out = rng.standard_normal((V, nout)).astype(np.float32)
    o_din, o_dW, _ = O.conv_bwd(x, d_out, W, rb)
    r_din, r_dW = np.zeros_like(x), np.zeros_like(W)
    ref.ref_rule_conv_bwd(x, r_din, V, nin, d_out, V, nout, W, r_dW, rb.rules, rb.counts, rb.vol, rb.cap, 0)
    assert _rel(o_din, r_din) < 1e-5
    assert _rel(o_dW, r_dW) < 1e-5


@pytest.mark.parametrize("size,stride", [((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((1, 1, 8), (1, 1, 1))])
def test_strided_conv_and_deconv_match_reference_kernels(ref, size, stride):
    il = _scene(3000, 10)
    V = il["V"]
    sp_in = np.array([64, 64, 8]) if size == (1, 1, 8) else np.array(S.FULL_SCALE)
    coords = il["coords"].copy()
    if size == (1, 1, 8):
        coords[:, :3] = coords[:, :3] % sp_in
        _, uniq = np.unique(coords, axis=0, return_index=True)
        coords = coords[np.sort(uniq)]
        V = coords.shape[0]
    out_sp = (sp_in - np.array(size)) // np.array(stride) + 1
    rb, out_coords = O.convolution_rules(coords, size, stride, out_sp)
    Vo = out_coords.shape[0]
    nin, nout = 32, 48
    rng = np.random.default_rng(11)
    x = rng.standard_normal((V, nin)).astype(np.float32)
    W = (rng.standard_normal((rb.vol, nin, nout)) * 0.1).astype(np.float32)
    # Convolution: in rows = column 0, out rows = column 1
    o_out, _ = O.conv_fwd(x, W, rb, Vo)
    r_out = np.zeros((Vo, nout), np.float32)
    ref.ref_rule_conv_fwd(x, V, nin, r_out, Vo, nout, W, rb.rules, rb.counts, rb.vol, rb.cap, 0)
    assert _rel(o_out, r_out) < 1e-5
    d_out = rng.standard_normal((Vo, nout)).astype(np.float32)
    o_din, o_dW, _ = O.conv_bwd(x, d_out, W, rb)
    r_din, r_dW = np.zeros_like(x), np.zeros_like(W)
    ref.ref_rule_conv_bwd(x, r_din, V, nin, d_out, Vo, nout, W, r_dW, rb.rules, rb.counts, rb.vol, rb.cap, 0)
    assert _rel(o_din, r_din) < 1e-5 and _rel(o_dW, r_dW) < 1e-5
    # Deconvolution (CPU/Deconvolution.cpp:15-16): same book, columns swapped: coarse rows in, fine rows out
    y = rng.standard_normal((Vo, nin)).astype(np.float32)
    o_up, _ = O.conv_fwd(y, W, rb, V, in_col=1)
    r_up = np.zeros((V, nout), np.float32)
    ref.ref_rule_conv_fwd(y, Vo, nin, r_up, V, nout, W, rb.rules, rb.counts, rb.vol, rb.cap, 1)
    assert _rel(o_up, r_up) < 1e-5
    d_up = rng.standard_normal((V, nout)).astype(np.float32)
    o_dy, o_dW2, _ = O.conv_bwd(y, d_up, W, rb, in_col=1)
    r_dy, r_dW2 = np.zeros_like(y), np.zeros_like(W)
    ref.ref_rule_conv_bwd(y, r_dy, Vo, nin, d_up, V, nout, W, r_dW2, rb.rules, rb.counts, rb.vol, rb.cap, 1)
    assert _rel(o_dy, r_dy) < 1e-5 and _rel(o_dW2, r_dW2) < 1e-5


# ------------------------------------------------------------------ SparseToDense: exact
def test_sparse_to_dense_matches_reference_kernels(ref):
    rng = np.random.default_rng(2)
    sp = np.array([12, 10, 6], np.int64)
    B, planes = 3, 7
    pts = np.unique(np.stack([rng.integers(0, sp[0], 400), rng.integers(0, sp[1], 400), rng.integers(0, sp[2], 400),
                              np.sort(rng.integers(0, B, 400))], 1), axis=0)
    pts = pts[np.argsort(pts[:, 3], kind="stable")]
    feats = rng.standard_normal((pts.shape[0], planes)).astype(np.float32)
    o_dense = O.sparse_to_dense(pts, feats, sp, B)
    vol = int(sp.prod())
    r_dense = np.zeros_like(o_dense)
    d_dense = rng.standard_normal(o_dense.shape).astype(np.float32)
    r_din = np.zeros_like(feats)
    for b in range(B):
        rows = np.nonzero(pts[:, 3] == b)[0]
        # SparseToDense_InputSgToRules (ConvolutionRules.h:109-123): (row, linearised offset, last dim fastest)
        lin = (pts[rows, 0] * sp[1] + pts[rows, 1]) * sp[2] + pts[rows, 2]
        rules = np.ascontiguousarray(np.stack([rows, lin], 1).astype(np.int32))
        ref.ref_sparse_to_dense_fwd(feats, r_dense[b].ctypes.data_as(vp), planes, vol, rules, len(rows))
        ref.ref_sparse_to_dense_bwd(r_din, d_dense[b].ctypes.data_as(vp), planes, vol, rules, len(rows))
    assert np.array_equal(o_dense, r_dense)
    assert np.array_equal(O.sparse_to_dense_bwd(pts, d_dense, planes, sp), r_din)

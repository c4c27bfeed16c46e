"""The wide-layer convolution kernel (csrc/conv_wide.hip, k_conv_cs) against the oracle.  The library's own
dispatch only picks it for launches of >= 320 workgroups; here the library's CONV_WIDE knob (aabr_set_knob; the
environment is read once per process, never on a launch path) forces it for every supported shape, so tile edges, single-tile rule books, odd pair counts, bias,
strided / transposed tables and several channel groups are all exercised at sizes the oracle finishes in seconds.
One case runs at a size where the default dispatch takes the wide path by itself."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scn():
    import sparseconvnet as scn
    return scn


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture
def force_wide():
    import _hip
    _hip.set_knob("CONV_WIDE", 1)
    yield
    _hip.set_knob("CONV_WIDE", None)


def _scene(rng, n, size, batch, C):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64), rng.standard_normal((n, C)).astype(np.float32)


def _variant():
    import _hip
    return _hip.load().aabr_conv_last_variant().decode()


@pytest.mark.parametrize("nIn,nOut,npts,bias", [(32, 64, 700, False), (64, 64, 3000, True), (96, 128, 2500, False),
                                                (128, 128, 127, False), (128, 64, 129, True), (256, 192, 2000, False),
                                                (384, 64, 900, False)])
def test_wide_submanifold_forward_backward(force_wide, nIn, nOut, npts, bias):
    scn = _scn()
    rng = np.random.default_rng(nIn * 7 + nOut + npts)
    coords, feats = _scene(rng, npts, (12, 11, 5), 2, nIn)
    f = _t(feats).requires_grad_(True)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), f])
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, bias).to(DEV)
    if bias:
        conv.bias.data.normal_()
    y = conv(x)
    assert _variant().startswith("k_conv_cs<"), _variant()
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = conv.weight.detach().cpu().numpy().reshape(27, nIn, nOut)
    b = conv.bias.detach().cpu().numpy() if bias else None
    ref, _ = O.conv_fwd(il["out"], W, rb, il["V"], b)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max() * nIn)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, db = O.conv_bwd(il["out"], g, W, rb, want_bias=bias)
    d_feats = O.input_layer_bwd(il, d_in)
    np.testing.assert_allclose(f.grad.cpu().numpy(), d_feats, rtol=1e-4, atol=2e-6 * np.abs(d_feats).max() * nOut)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(27, nIn, nOut), dW, rtol=1e-4,
                               atol=1e-5 * np.abs(dW).max())
    # bit-reproducible: same launch, same bits
    with torch.no_grad():
        assert torch.equal(conv(x).features, conv(x).features)


@pytest.mark.parametrize("fs,st,nIn,nOut", [([2, 2, 2], [2, 2, 2], 64, 128), ([1, 1, 8], [1, 1, 1], 128, 128),
                                            ([3, 3, 3], [2, 2, 2], 32, 64)])
def test_wide_strided_conv_and_deconv(force_wide, fs, st, nIn, nOut):
    scn = _scn()
    rng = np.random.default_rng(90 + nIn)
    size = np.array([16, 16, 8]) if fs != [3, 3, 3] else np.array([17, 17, 9])
    coords, feats = _scene(rng, 2500, tuple(size), 2, nIn)
    f = _t(feats).requires_grad_(True)
    x = scn.InputLayer(3, list(size), mode=4)([_t(coords), f])
    conv = scn.Convolution(3, nIn, nOut, fs, st, False).to(DEV)
    dec = scn.Deconvolution(3, nOut, nIn - nIn % 64 if nIn % 64 == 0 else 64, fs, st, False).to(DEV)
    nBack = dec.weight.shape[3]
    y = conv(x)
    assert _variant().startswith("k_conv_cs<"), _variant()
    z = dec(y)
    assert _variant().startswith("k_conv_cs<"), _variant()
    il = O.input_layer(coords, feats, 4)
    osz = (size - np.array(fs)) // np.array(st) + 1
    rb, oc = O.convolution_rules(il["coords"], fs, st, osz)
    Wc = conv.weight.detach().cpu().numpy().reshape(rb.vol, nIn, nOut)
    Wd = dec.weight.detach().cpu().numpy().reshape(rb.vol, nOut, nBack)
    yr, _ = O.conv_fwd(il["out"], Wc, rb, oc.shape[0])
    zr, _ = O.conv_fwd(yr, Wd, rb, il["V"], in_col=1)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), yr, rtol=1e-4, atol=1e-5 * np.abs(yr).max())
    np.testing.assert_allclose(z.features.detach().cpu().numpy(), zr, rtol=1e-4, atol=1e-5 * np.abs(zr).max())
    g = rng.standard_normal(zr.shape).astype(np.float32)
    z.features.backward(_t(g))
    d_y, dWd, _ = O.conv_bwd(yr, g, Wd, rb, in_col=1)
    d_x, dWc, _ = O.conv_bwd(il["out"], d_y, Wc, rb)
    np.testing.assert_allclose(dec.weight.grad.cpu().numpy().reshape(Wd.shape), dWd, rtol=1e-4,
                               atol=1e-5 * np.abs(dWd).max())
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(Wc.shape), dWc, rtol=1e-4,
                               atol=1e-5 * np.abs(dWc).max())
    d_feats = O.input_layer_bwd(il, d_x)
    np.testing.assert_allclose(f.grad.cpu().numpy(), d_feats, rtol=1e-4, atol=1e-5 * np.abs(d_feats).max())


def test_default_dispatch_takes_the_wide_kernel_on_a_large_layer():
    """no env knob: 60k sites x 128 output planes = 940 workgroups -> the library picks k_conv_cs by itself"""
    scn = _scn()
    import synth_scenes as S
    locs, feats = S.make_batch(1, 80000, 3, 20)
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((locs.shape[0], 64)).astype(np.float32)
    x = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)([_t(locs), _t(feats)])
    conv = scn.SubmanifoldConvolution(3, 64, 128, 3, False).to(DEV)
    with torch.no_grad():
        y = conv(x)
    assert _variant().startswith("k_conv_cs<2,0,"), _variant()
    il = O.input_layer(locs, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    ref, _ = O.conv_fwd(il["out"], conv.weight.detach().cpu().numpy().reshape(27, 64, 128), rb, il["V"])
    np.testing.assert_allclose(y.features.cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max() * 64)


@pytest.mark.parametrize("nIn,nOut,npts,vol3", [(64, 64, 3000, True), (128, 128, 2500, True), (128, 64, 129, True),
                                                (192, 128, 1500, True), (256, 192, 2000, True), (512, 64, 700, True),
                                                (128, 128, 900, False)])
def test_wide_bf16_storage_matches_oracle_on_rounded_operands(nIn, nOut, npts, vol3):
    """k_conv_cs<.., bf16> through the C ABI (forced for every supported shape) against the oracle fed with the SAME
    bf16-rounded features and weights: the products are exact in fp32 either way, so the only differences are the
    fp32 accumulation order and the final rounding to bf16 (<= 1 bf16 ulp of the largest term); forward form and
    input-gradient form (transposed pack, mirrored offsets), 1..4 chunks per row and two channel groups (512)."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + 3 * nOut + npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    fsz = [3, 3, 3] if vol3 else [1, 1, 1]
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor(fsz))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], fsz)
    T = lib.aabr_conv_wide_tile_rows_bf16(nIn, nOut, V, V, vol) or 128
    _hip.set_knob("CONV_WIDE_BF16", 1)
    try:
        T = lib.aabr_conv_wide_tile_rows_bf16(nIn, nOut, V, V, vol)
    finally:
        _hip.set_knob("CONV_WIDE_BF16", None)
    assert T in (64, 80, 96, 112, 128)
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    blocks = ga.blocks_wide(T)
    # forward form
    f = torch.as_tensor(rng.standard_normal((V, nIn)).astype(np.float32)).to(DEV).bfloat16()
    out = torch.empty((V, nOut), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_wide_bf16(ptr(f), nIn, V, ptr(out), nOut, V, ptr(blocks), T, vol, None, 0, ptr(pf),
                                          stream()))
    assert ",bf16" in _variant(), _variant()
    ref, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, V, None)
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).max())
    # input-gradient form: d_in = sum_k d_out[partner] @ W[k]^T over the mirrored offsets
    g = torch.as_tensor(rng.standard_normal((V, nOut)).astype(np.float32)).to(DEV).bfloat16()
    d_in = torch.empty((V, nIn), dtype=torch.bfloat16, device=DEV)
    Tb = T
    _hip.set_knob("CONV_WIDE_BF16", 1)
    try:
        Tb = lib.aabr_conv_wide_tile_rows_bf16(nOut, nIn, V, V, vol)
    finally:
        _hip.set_knob("CONV_WIDE_BF16", None)
    if Tb:
        check(lib.aabr_conv_forward_wide_bf16(ptr(g), nOut, V, ptr(d_in), nIn, V, ptr(ga.blocks_wide(Tb)), Tb, vol,
                                              None, 1 | 2, ptr(pt), stream()))
        dref, _, _ = O.conv_bwd(np.zeros((V, nIn), np.float32), g.float().cpu().numpy(), Wr, rb, want_bias=False)
        np.testing.assert_allclose(d_in.float().cpu().numpy(), dref, rtol=2 ** -7, atol=2 ** -7 * np.abs(dref).max())
    # same launch, same bits
    out2 = torch.empty_like(out)
    check(lib.aabr_conv_forward_wide_bf16(ptr(f), nIn, V, ptr(out2), nOut, V, ptr(blocks), T, vol, None, 0, ptr(pf),
                                          stream()))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("nIn,nOut", [(64, 128), (128, 64)])
def test_wide_backward_pybind_api_without_prepack(force_wide, nIn, nOut):
    """`SCN.SubmanifoldConvolution_backward` called the pybind way (pack_t=None, no
    WeightPackPlan) on a NON-SQUARE wide-eligible layer packs its own transposed weights; the pack call must get
    the launch's (n_in, n_out) = (nOut, nIn), not the weight's (size(2), size(3))."""
    from sparseconvnet import SCN
    rng = np.random.default_rng(nIn + 3 * nOut)
    coords, feats = _scene(rng, 1500, (12, 11, 5), 2, nIn)
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    md = SCN.Metadata_3()
    sz, fs = torch.LongTensor([16, 16, 8]), torch.LongTensor([3, 3, 3])
    x = torch.empty(0, device=DEV)
    SCN.InputLayer_updateOutput(md, sz, _t(coords), _t(feats), x, 2, 4)
    W = (rng.standard_normal((27, 1, nIn, nOut)) * 0.1).astype(np.float32)
    g = rng.standard_normal((il["V"], nOut)).astype(np.float32)
    d_in, d_w = torch.empty(0, device=DEV), torch.zeros((27, 1, nIn, nOut), device=DEV)
    SCN.SubmanifoldConvolution_backward(sz, fs, md, x, d_in, _t(g), _t(W), d_w, torch.empty(0, device=DEV))
    assert SCN.pack_stats["own"] > 0
    want_in, want_w, _ = O.conv_bwd(il["out"], g, W.reshape(27, nIn, nOut), rb)
    np.testing.assert_allclose(d_in.cpu().numpy(), want_in, rtol=1e-4, atol=2e-6 * np.abs(want_in).max() * nOut)
    np.testing.assert_allclose(d_w.cpu().numpy().reshape(27, nIn, nOut), want_w, rtol=1e-4,
                               atol=1e-5 * np.abs(want_w).max())


@pytest.mark.parametrize("bf,nIn,nOut,npts,use_res", [(False, 64, 64, 3000, False), (False, 128, 128, 2500, True),
                                                      (False, 32, 192, 700, False), (True, 64, 128, 2500, False),
                                                      (True, 128, 64, 129, False)])
def test_wide_write_out_statistics_feed_batchnorm(bf, nIn, nOut, npts, use_res):
    """aabr_conv_forward_wide[_bf16]_stats: the per-tile fp64 column sums of the write-out equal the sums of the
    STORED output (numpy fp64, same tile grouping: <= 1 ulp of fp64 apart), leave the output bits unchanged, and
    aabr_bn_forward_parts fed with them reproduces aabr_bn_forward (which makes its own statistics pass) to fp32
    rounding of the same statistics (SCN/CPU/BatchNormalization.cpp:20-48)."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + 5 * nOut + npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    knob = "CONV_WIDE_BF16" if bf else "CONV_WIDE"
    _hip.set_knob(knob, 1)
    try:
        T = (lib.aabr_conv_wide_tile_rows_bf16 if bf else lib.aabr_conv_wide_tile_rows)(nIn, nOut, V, V, vol)
    finally:
        _hip.set_knob(knob, None)
    assert T >= 64
    ntile = (V + T - 1) // T
    assert lib.aabr_conv_wide_stats_doubles(V, T, nOut) == ntile * 2 * nOut
    W = _t((rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32))
    blocks = ga.blocks_wide(T)
    dt = torch.bfloat16 if bf else torch.float32
    f = _t(rng.standard_normal((V, nIn)).astype(np.float32)).to(dt)
    out, out0 = torch.empty((V, nOut), dtype=dt, device=DEV), torch.empty((V, nOut), dtype=dt, device=DEV)
    stats = torch.full((ntile, 2, nOut), float("nan"), dtype=torch.float64, device=DEV)
    bias = _t(rng.standard_normal(nOut).astype(np.float32))
    res = _t(rng.standard_normal((V, nOut)).astype(np.float32)) if use_res else None
    if bf:
        n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
        pf, pt = (torch.empty(n, dtype=dt, device=DEV) for _ in range(2))
        check(lib.aabr_conv_pack_weights2_bf16(ptr(W), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
        args = (ptr(f), nIn, V, None, nOut, V, ptr(blocks), T, vol, ptr(bias), 0, ptr(pf))
        check(lib.aabr_conv_forward_wide_bf16_stats(*args[:3], ptr(out), *args[4:], ptr(stats), stream()))
        check(lib.aabr_conv_forward_wide_bf16(*args[:3], ptr(out0), *args[4:], stream()))
    else:
        wp = torch.empty(lib.aabr_conv_wpack_floats(vol, nIn, nOut), device=DEV)
        check(lib.aabr_conv_pack_weights(ptr(W), vol, nIn, nOut, 0, ptr(wp), stream()))
        args = (ptr(f), nIn, V, None, nOut, V, ptr(blocks), T, vol, ptr(bias), 0, ptr(wp), ptr(res) if use_res else None)
        check(lib.aabr_conv_forward_wide_stats(*args[:3], ptr(out), *args[4:], ptr(stats), stream()))
        check(lib.aabr_conv_forward_wide_res(*args[:3], ptr(out0), *args[4:], stream()))
    assert torch.equal(out, out0)                      # the statistics do not touch the output
    o64 = out.double().cpu().numpy()
    st = stats.cpu().numpy()
    assert np.isfinite(st).all()
    for j in range(ntile):
        blk = o64[j * T:(j + 1) * T]
        np.testing.assert_allclose(st[j, 0], blk.sum(0), rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(st[j, 1], (blk * blk).sum(0), rtol=1e-13, atol=1e-12)
    # BatchNorm from the partials == BatchNorm with its own statistics pass
    ws = torch.empty(int(lib.aabr_bn_scratch_floats(nOut)), device=DEV)
    gam, bet = _t(rng.uniform(0.5, 1.5, nOut).astype(np.float32)), _t(rng.standard_normal(nOut).astype(np.float32))

    def run(parts):
        y = torch.empty_like(out)
        sm, si, rm, rv = (torch.zeros(nOut, device=DEV) for _ in range(4))
        rv.fill_(1.0)
        a = (ptr(out), ptr(y), V, nOut, ptr(sm), ptr(si), ptr(rm), ptr(rv), ptr(gam), ptr(bet), 1e-4, 0.9)
        if parts:
            fn = lib.aabr_bn_forward_parts_bf16 if bf else lib.aabr_bn_forward_parts
            check(fn(*a, 0.2, ptr(stats), ntile, ptr(ws), stream()))
        else:
            _hip.set_knob("BN_SMALL", 0)   # the three-launch path, whatever the row count
            try:
                fn = lib.aabr_bn_forward_bf16 if bf else lib.aabr_bn_forward
                check(fn(*a, 1, 0.2, ptr(ws), stream()))
            finally:
                _hip.set_knob("BN_SMALL", None)
        return [t.float().cpu().numpy() for t in (y, sm, si, rm, rv)]

    got, want = run(True), run(False)
    for g_, w_ in zip(got[1:], want[1:]):
        np.testing.assert_allclose(g_, w_, rtol=3e-7, atol=1e-9)       # the same sums up to the last fp64 bits
    np.testing.assert_allclose(got[0], want[0], rtol=2 ** -7 if bf else 1e-6, atol=2 ** -7 if bf else 1e-6)


@pytest.mark.parametrize("nIn,nOut,npts,use_res,leak,affine", [(64, 64, 3000, False, 0.0, True), (128, 128, 2500, True, 0.2, True),
                                                               (128, 64, 900, False, 0.0, False)])
def test_wide_write_out_backward_statistics_feed_batchnorm_backward(force_wide, nIn, nOut, npts, use_res, leak, affine):
    """aabr_conv_forward_wide_bwd_stats (the input-gradient launch of the layer that consumed a BatchNorm's output):
    its per-tile partial sums equal numpy's fp64 sums of the masked d_out it stores, and aabr_bn_backward_parts fed
    with them gives the d_in / d_weight / d_bias of aabr_bn_backward_add (own statistics pass) to the last bits of the
    fp64 sums (SCN/CPU/BatchNormalization.cpp:66-106)."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + 9 * nOut + npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    T = lib.aabr_conv_wide_tile_rows(nIn, nOut, V, V, vol)
    assert T >= 64
    ntile = (V + T - 1) // T
    # the conv launch: g [V, nIn] -> d_out [V, nOut] (any wide launch will do; the transposed flags are exercised elsewhere)
    W = _t((rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32))
    wp = torch.empty(lib.aabr_conv_wpack_floats(vol, nIn, nOut), device=DEV)
    check(lib.aabr_conv_pack_weights(ptr(W), vol, nIn, nOut, 0, ptr(wp), stream()))
    g = _t(rng.standard_normal((V, nIn)).astype(np.float32))
    res = _t(rng.standard_normal((V, nOut)).astype(np.float32)) if use_res else None
    # the BatchNorm whose output the layer consumed: input xb, saved statistics, affine parameters
    xb = _t((rng.standard_normal((V, nOut)) * 1.3 + 0.2).astype(np.float32))
    gam = _t(rng.uniform(0.5, 1.5, nOut).astype(np.float32)) if affine else None
    bet = _t(rng.standard_normal(nOut).astype(np.float32)) if affine else None
    ws = torch.empty(int(lib.aabr_bn_scratch_floats(nOut)), device=DEV)
    y = torch.empty_like(xb)
    sm, si, rm, rv = (torch.zeros(nOut, device=DEV) for _ in range(4))
    _hip.set_knob("BN_SMALL", 0)
    try:
        check(lib.aabr_bn_forward(ptr(xb), ptr(y), V, nOut, ptr(sm), ptr(si), ptr(rm), ptr(rv), ptr(gam) if affine else None,
                                  ptr(bet) if affine else None, 1e-4, 0.9, 1, leak, ptr(ws), stream()))
        d_out = torch.empty((V, nOut), device=DEV)
        d_out0 = torch.empty_like(d_out)
        stats = torch.full((ntile, 2, nOut), float("nan"), dtype=torch.float64, device=DEV)
        blocks = ga.blocks_wide(T)
        a = (ptr(g), nIn, V, None, nOut, V, ptr(blocks), T, vol, None, 0, ptr(wp), ptr(res) if use_res else None)
        check(lib.aabr_conv_forward_wide_bwd_stats(*a[:3], ptr(d_out), *a[4:], ptr(stats), ptr(xb), ptr(sm), ptr(si),
                                                   ptr(gam) if affine else None, ptr(bet) if affine else None, leak, stream()))
        check(lib.aabr_conv_forward_wide_res(*a[:3], ptr(d_out0), *a[4:], stream()))
        assert torch.equal(d_out, d_out0)
        # numpy fp64 sums of the masked gradient per tile
        d32 = d_out.cpu().numpy()
        mask = (y.cpu().numpy() > 0)
        dm = np.where(mask, d32, d32 * np.float32(leak)).astype(np.float64)     # the mask multiply is an fp32 operation
        xc = (xb.cpu().numpy().astype(np.float32) - sm.cpu().numpy().astype(np.float32)).astype(np.float64)
        st = stats.cpu().numpy()
        for j in range(ntile):
            sl = slice(j * T, (j + 1) * T)
            np.testing.assert_allclose(st[j, 0], dm[sl].sum(0), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(st[j, 1], (xc[sl] * dm[sl]).sum(0), rtol=1e-12, atol=1e-11)
        radd = _t(rng.standard_normal((V, nOut)).astype(np.float32))

        def run(parts):
            d_in = torch.empty_like(xb)
            dw, db = torch.zeros(nOut, device=DEV), torch.zeros(nOut, device=DEV)
            common = (ptr(xb), ptr(d_in), ptr(y), ptr(d_out), V, nOut, ptr(sm), ptr(si), ptr(gam) if affine else None,
                      ptr(bet) if affine else None, ptr(dw), ptr(db), leak)
            if parts:
                check(lib.aabr_bn_backward_parts(*common, ptr(stats), ntile, ptr(ws), ptr(radd), stream()))
            else:
                check(lib.aabr_bn_backward_add(*common, ptr(ws), ptr(radd), stream()))
            return d_in.cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()

        got, want = run(True), run(False)
        np.testing.assert_allclose(got[2], want[2], rtol=3e-7, atol=1e-6)
        np.testing.assert_allclose(got[1], want[1], rtol=3e-7, atol=1e-6)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-6, atol=1e-6)
    finally:
        _hip.set_knob("BN_SMALL", None)


@pytest.mark.parametrize("nIn,nOut,npts,parts", [(128, 128, 700, 0), (256, 256, 600, 0), (128, 64, 600, 0), (64, 128, 1500, 5),
                                                 (256, 128, 700, 27)])
def test_wide_offset_split_matches_oracle_and_unsplit(request, nIn, nOut, npts, parts):
    """aabr_conv_forward_wide_split (coarse maps: every (tile, slab) item cut into parts over the filter offsets,
    partial tiles summed in part order): forward, transposed (input-gradient form), with bias and residual, against the
    oracle (SCN/CPU/Convolution.cpp:117-185) and against the 64-row-tile kernel; the same call twice gives the same
    bits; the decision function picks it exactly where the wide kernel declines for lack of workgroups."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn * 3 + nOut + npts)
    coords, _ = _scene(rng, npts, (9, 8, 4), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    assert lib.aabr_conv_wide_tile_rows(nIn, nOut, V, V, vol) == 0          # too few workgroups for the unsplit form
    _hip.set_knob("SPLIT_MIN_ITEMS", 1)       # (below 8 (tile, slab) items the dispatch prefers the 16-column item kernel)
    request.addfinalizer(lambda: _hip.set_knob("SPLIT_MIN_ITEMS", None))
    v = lib.aabr_conv_wide_split(nIn, nOut, V, V, vol)
    assert v, "the split form should take this launch"
    T, P = v & 0xffff, v >> 16
    assert T == 64 and 2 <= P <= vol and (P == vol or ((V + 63) // 64) * (nOut // 64) * P >= 512)
    if parts:
        P = parts
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    f = rng.standard_normal((V, nIn)).astype(np.float32)
    b = rng.standard_normal(nOut).astype(np.float32)
    r = rng.standard_normal((V, nOut)).astype(np.float32)
    Wd, fd, bd, rd = _t(W), _t(f), _t(b), _t(r)
    wp = torch.empty(lib.aabr_conv_wpack_floats(vol, nIn, nOut), device=DEV)
    check(lib.aabr_conv_pack_weights(ptr(Wd), vol, nIn, nOut, 0, ptr(wp), stream()))
    scratch = torch.empty(int(lib.aabr_conv_wide_split_scratch_floats(V, nOut, P)), device=DEV)
    blocks = ga.blocks_wide(T)
    out = torch.empty((V, nOut), device=DEV)
    check(lib.aabr_conv_forward_wide_split(ptr(fd), nIn, V, ptr(out), nOut, V, ptr(blocks), T, vol, ptr(bd), 0, ptr(wp),
                                           ptr(rd), P, ptr(scratch), stream()))
    ref, _ = O.conv_fwd(f, W.reshape(vol, nIn, nOut), rb, V, b)
    ref = ref + r
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6 * np.abs(f).max() * nIn)
    out2 = torch.empty_like(out)
    scratch.fill_(float("nan"))                                              # every part writes its whole partial tile
    check(lib.aabr_conv_forward_wide_split(ptr(fd), nIn, V, ptr(out2), nOut, V, ptr(blocks), T, vol, ptr(bd), 0, ptr(wp),
                                           ptr(rd), P, ptr(scratch), stream()))
    assert torch.equal(out, out2)
    # against the 64-row-tile kernel (other summation order over the offsets: fp32 rounding apart)
    out3 = torch.empty_like(out)
    wp3 = torch.empty_like(wp)
    check(lib.aabr_conv_forward(ptr(fd), nIn, V, ptr(out3), nOut, V, ptr(ga.blocks()), vol, ptr(Wd), ptr(bd), 0, ptr(wp3),
                                stream()))
    np.testing.assert_allclose(got, (out3 + rd).cpu().numpy(), rtol=2e-5, atol=1e-5 * np.abs(ref).max())
    # transposed form: d_in = d_out @ W[k]^T over the mirrored offsets
    g = rng.standard_normal((V, nOut)).astype(np.float32)
    wt = torch.empty(lib.aabr_conv_wpack_floats(vol, nOut, nIn), device=DEV)
    check(lib.aabr_conv_pack_weights(ptr(Wd), vol, nOut, nIn, 1, ptr(wt), stream()))
    vb = lib.aabr_conv_wide_split(nOut, nIn, V, V, vol)
    assert vb
    Tb, Pb = vb & 0xffff, vb >> 16
    d_in = torch.empty((V, nIn), device=DEV)
    sc2 = torch.empty(int(lib.aabr_conv_wide_split_scratch_floats(V, nIn, Pb)), device=DEV)
    check(lib.aabr_conv_forward_wide_split(ptr(_t(g)), nOut, V, ptr(d_in), nIn, V, ptr(ga.blocks_wide(Tb)), Tb, vol, None, 3,
                                           ptr(wt), None, Pb, ptr(sc2), stream()))
    want, _, _ = O.conv_bwd(f, g, W.reshape(vol, nIn, nOut), rb)
    np.testing.assert_allclose(d_in.cpu().numpy(), want, rtol=1e-4, atol=2e-6 * np.abs(g).max() * nOut)
    # through the layer: the module dispatches the split form here
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, False).to(DEV)
    conv.weight.data.copy_(Wd)
    xs = scn.SparseConvNetTensor()
    xs.metadata, xs.spatial_size = x.metadata, x.spatial_size
    xs.features = fd.clone().requires_grad_(True)
    y = conv(xs)
    assert lib.aabr_conv_last_variant().decode().endswith("split>")
    ref0, _ = O.conv_fwd(f, W.reshape(vol, nIn, nOut), rb, V)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref0, rtol=1e-4, atol=2e-6 * np.abs(f).max() * nIn)
    y.features.backward(_t(g))
    np.testing.assert_allclose(xs.features.grad.cpu().numpy(), want, rtol=1e-4, atol=2e-6 * np.abs(g).max() * nOut)


@pytest.mark.parametrize("nIn,nOut,npts,parts", [(128, 128, 700, 0), (256, 256, 600, 0), (128, 64, 600, 0),
                                                 (64, 128, 1500, 5), (256, 128, 700, 27), (512, 64, 400, 0)])
def test_wide_offset_split_bf16_storage(request, nIn, nOut, npts, parts):
    """aabr_conv_forward_wide_split_bf16 (the offset split for a bf16-storage pass's coarse scales): forward and
    input-gradient form against the oracle (SCN/CPU/Convolution.cpp:117-185) on the SAME bf16-rounded features and
    weights -- products exact in fp32, differences = accumulation order + ONE rounding to bf16 in the reduce kernel
    (tolerance 2^-7 of the largest term, as for the unsplit bf16 kernel); same call twice gives the same bits; the
    layer under `.to(bfloat16)` dispatches it."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn * 5 + nOut + npts)
    coords, _ = _scene(rng, npts, (9, 8, 4), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    assert lib.aabr_conv_wide_tile_rows_bf16(nIn, nOut, V, V, vol) == 0
    _hip.set_knob("SPLIT_MIN_ITEMS", 1)
    request.addfinalizer(lambda: _hip.set_knob("SPLIT_MIN_ITEMS", None))
    v = lib.aabr_conv_wide_split_bf16(nIn, nOut, V, V, vol)
    assert v, "the split form should take this launch"
    T, P = v & 0xffff, v >> 16
    assert T == 64 and 2 <= P <= vol
    if parts:
        P = parts
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    f = torch.as_tensor(rng.standard_normal((V, nIn)).astype(np.float32)).to(DEV).bfloat16()
    b = rng.standard_normal(nOut).astype(np.float32)
    out = torch.empty((V, nOut), dtype=torch.bfloat16, device=DEV)
    scratch = torch.full((int(lib.aabr_conv_wide_split_scratch_floats(V, nOut, P)),), float("nan"), device=DEV)
    blocks = ga.blocks_wide(T)
    check(lib.aabr_conv_forward_wide_split_bf16(ptr(f), nIn, V, ptr(out), nOut, V, ptr(blocks), T, vol, ptr(_t(b)), 0,
                                                ptr(pf), P, ptr(scratch), stream()))
    assert _variant().endswith("bf16,split>"), _variant()
    ref, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, V, b)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).max())
    out2 = torch.empty_like(out)
    check(lib.aabr_conv_forward_wide_split_bf16(ptr(f), nIn, V, ptr(out2), nOut, V, ptr(blocks), T, vol, ptr(_t(b)), 0,
                                                ptr(pf), P, ptr(scratch), stream()))
    assert torch.equal(out, out2)
    # input-gradient form
    g = torch.as_tensor(rng.standard_normal((V, nOut)).astype(np.float32)).to(DEV).bfloat16()
    vb = lib.aabr_conv_wide_split_bf16(nOut, nIn, V, V, vol)
    assert vb
    Tb, Pb = vb & 0xffff, vb >> 16
    d_in = torch.empty((V, nIn), dtype=torch.bfloat16, device=DEV)
    sc2 = torch.empty(int(lib.aabr_conv_wide_split_scratch_floats(V, nIn, Pb)), device=DEV)
    check(lib.aabr_conv_forward_wide_split_bf16(ptr(g), nOut, V, ptr(d_in), nIn, V, ptr(ga.blocks_wide(Tb)), Tb, vol, None,
                                                3, ptr(pt), Pb, ptr(sc2), stream()))
    dref, _, _ = O.conv_bwd(np.zeros((V, nIn), np.float32), g.float().cpu().numpy(), Wr, rb, want_bias=False)
    np.testing.assert_allclose(d_in.float().cpu().numpy(), dref, rtol=2 ** -7, atol=2 ** -7 * np.abs(dref).max())
    # through the layer in bf16 storage
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, False).to(DEV)
    conv.weight.data.copy_(Wd)
    xs = scn.SparseConvNetTensor()
    xs.metadata, xs.spatial_size = x.metadata, x.spatial_size
    xs.features = f.clone().requires_grad_(True)
    y = conv(xs)
    assert y.features.dtype == torch.bfloat16
    if _variant().endswith("bf16,split>"):            # (the layer prepacks in bf16 storage; otherwise the item kernel ran)
        ref0, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, V)
        np.testing.assert_allclose(y.features.detach().float().cpu().numpy(), ref0, rtol=2 ** -7,
                                   atol=2 ** -7 * np.abs(ref0).max())
        y.features.backward(g)
        np.testing.assert_allclose(xs.features.grad.float().cpu().numpy(), dref, rtol=2 ** -7,
                                   atol=2 ** -7 * np.abs(dref).max())
    else:
        pytest.fail("the bf16 layer did not dispatch the split form: " + _variant())


@pytest.mark.parametrize("nIn,nOut,npts,leak", [(64, 64, 3000, 0.0), (128, 128, 2500, 0.2), (128, 64, 900, 0.0),
                                                (64, 128, 2500, 0.0)])
def test_wide_bf16_write_out_backward_statistics_feed_batchnorm_backward(nIn, nOut, npts, leak):
    """aabr_conv_forward_wide_bf16_bwd_stats: per-tile fp64 sums of the STORED (bf16) masked d_out and of (x - mean) *
    d, the mask from the BatchNorm's stored output (SCN/CPU/BatchNormalization.cpp:66-84 on the bf16-storage model);
    aabr_bn_backward_parts_bf16 fed with them returns what aabr_bn_backward_bf16 (own statistics pass) returns."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + 11 * nOut + npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    _hip.set_knob("CONV_WIDE_BF16", 1)
    _hip.set_knob("BN_SMALL", 0)
    try:
        T = lib.aabr_conv_wide_tile_rows_bf16(nIn, nOut, V, V, vol)
        assert T >= 64
        ntile = (V + T - 1) // T
        W = _t((rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32))
        n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
        pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        check(lib.aabr_conv_pack_weights2_bf16(ptr(W), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
        g = _t(rng.standard_normal((V, nIn)).astype(np.float32)).bfloat16()
        xb = _t((rng.standard_normal((V, nOut)) * 1.3 + 0.2).astype(np.float32)).bfloat16()
        gam = _t(rng.uniform(0.5, 1.5, nOut).astype(np.float32))
        bet = _t(rng.standard_normal(nOut).astype(np.float32))
        ws = torch.empty(int(lib.aabr_bn_scratch_floats(nOut)), device=DEV)
        y = torch.empty_like(xb)
        sm, si, rm, rv = (torch.zeros(nOut, device=DEV) for _ in range(4))
        check(lib.aabr_bn_forward_bf16(ptr(xb), ptr(y), V, nOut, ptr(sm), ptr(si), ptr(rm), ptr(rv), ptr(gam), ptr(bet),
                                       1e-4, 0.9, 1, leak, ptr(ws), stream()))
        d_out = torch.empty((V, nOut), dtype=torch.bfloat16, device=DEV)
        d_out0 = torch.empty_like(d_out)
        stats = torch.full((ntile, 2, nOut), float("nan"), dtype=torch.float64, device=DEV)
        blocks = ga.blocks_wide(T)
        a = (ptr(g), nIn, V, None, nOut, V, ptr(blocks), T, vol, None, 0, ptr(pf))
        check(lib.aabr_conv_forward_wide_bf16_bwd_stats(*a[:3], ptr(d_out), *a[4:], ptr(stats), ptr(xb), ptr(y), ptr(sm),
                                                        leak, stream()))
        check(lib.aabr_conv_forward_wide_bf16(*a[:3], ptr(d_out0), *a[4:], stream()))
        assert torch.equal(d_out, d_out0)
        d32 = d_out.float().cpu().numpy()
        mask = y.float().cpu().numpy() > 0
        dm = np.where(mask, d32, d32 * np.float32(leak)).astype(np.float64)
        xc = (xb.float().cpu().numpy() - sm.cpu().numpy().astype(np.float32)).astype(np.float64)
        st = stats.cpu().numpy()
        for j in range(ntile):
            sl = slice(j * T, (j + 1) * T)
            np.testing.assert_allclose(st[j, 0], dm[sl].sum(0), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(st[j, 1], (xc[sl] * dm[sl]).sum(0), rtol=1e-12, atol=1e-11)

        def run(parts):
            d_in = torch.empty_like(xb)
            dw, db = torch.zeros(nOut, device=DEV), torch.zeros(nOut, device=DEV)
            common = (ptr(xb), ptr(d_in), ptr(y), ptr(d_out), V, nOut, ptr(sm), ptr(si), ptr(gam), ptr(bet), ptr(dw),
                      ptr(db), leak)
            if parts:
                check(lib.aabr_bn_backward_parts_bf16(*common, ptr(stats), ntile, ptr(ws), stream()))
            else:
                check(lib.aabr_bn_backward_bf16(*common, ptr(ws), stream()))
            return d_in.float().cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()

        got, want = run(True), run(False)
        np.testing.assert_allclose(got[2], want[2], rtol=3e-7, atol=1e-6)
        np.testing.assert_allclose(got[1], want[1], rtol=3e-7, atol=1e-6)
        # d_in is stored in bf16: the fp64 sums agree to ~1e-16, so a stored value differs by at most one bf16 step, rarely
        diff = np.abs(got[0] - want[0])
        assert (diff > 0).mean() < 1e-3 and np.all(diff <= 2 ** -7 * np.abs(want[0]) + 1e-30)
    finally:
        _hip.set_knob("BN_SMALL", None)
        _hip.set_knob("CONV_WIDE_BF16", None)



@pytest.mark.parametrize("nIn,nOut,npts", [(64, 64, 3000), (128, 128, 2500), (128, 64, 900), (256, 128, 1200)])
def test_bf16_residual_in_the_write_out_equals_store_then_add(request, nIn, nOut, npts):
    """round 6: aabr_conv_forward_wide_bf16_res / aabr_conv_forward_wide_split_bf16_res / aabr_bn_backward_add_bf16 fold the
    consumer's bf16 add into the producing write-out.  The separate form stores bf16(conv) and then bf16(float(a) +
    float(b)) -- two roundings -- and the fused write-outs round their own value before the sum: the results must be EQUAL,
    bit for bit, to "unfused launch + aabr_add(bf16)" (which tests/test_gpu_plan.py holds equal to torch's bf16 add), also
    with the forward statistics of the stored values riding along."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn * 3 + nOut + npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    Wd = _t((rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32))
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    f = torch.as_tensor(rng.standard_normal((V, nIn)).astype(np.float32)).to(DEV).bfloat16()
    res = torch.as_tensor(rng.standard_normal((V, nOut)).astype(np.float32) * 3).to(DEV).bfloat16()
    # ---- wide kernel
    _hip.set_knob("CONV_WIDE_BF16", 1)
    request.addfinalizer(lambda: _hip.set_knob("CONV_WIDE_BF16", None))
    T = lib.aabr_conv_wide_tile_rows_bf16(nIn, nOut, V, V, vol)
    assert T >= 64
    blocks = ga.blocks_wide(T)
    plain = torch.empty((V, nOut), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_wide_bf16(ptr(f), nIn, V, ptr(plain), nOut, V, ptr(blocks), T, vol, None, 0, ptr(pf), stream()))
    want = torch.empty_like(plain)
    check(lib.aabr_add(ptr(plain), ptr(res), ptr(want), V * nOut, 1, stream()))
    assert torch.equal(want, plain + res)
    ntile = (V + T - 1) // T
    stats = torch.zeros(ntile * 2 * nOut, dtype=torch.float64, device=DEV)
    got = torch.empty_like(plain)
    check(lib.aabr_conv_forward_wide_bf16_res(ptr(f), nIn, V, ptr(got), nOut, V, ptr(blocks), T, vol, None, 0, ptr(pf),
                                              ptr(res), ptr(stats), None, None, None, 0.0, stream()))
    assert torch.equal(got, want)
    s = stats.view(ntile, 2, nOut).sum(0)                      # the statistics are those of the STORED sums
    np.testing.assert_allclose(s[0].cpu().numpy(), want.double().sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(s[1].cpu().numpy(), want.double().square().sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    # ---- offset split (second stage adds the residual)
    _hip.set_knob("SPLIT_MIN_ITEMS", 1)
    request.addfinalizer(lambda: _hip.set_knob("SPLIT_MIN_ITEMS", None))
    P, Ts = 5, 64
    scratch = torch.empty(int(lib.aabr_conv_wide_split_scratch_floats(V, nOut, P)), device=DEV)
    sp_plain = torch.empty_like(plain)
    check(lib.aabr_conv_forward_wide_split_bf16(ptr(f), nIn, V, ptr(sp_plain), nOut, V, ptr(ga.blocks_wide(Ts)), Ts, vol,
                                                None, 0, ptr(pf), P, ptr(scratch), stream()))
    sp_got = torch.empty_like(plain)
    check(lib.aabr_conv_forward_wide_split_bf16_res(ptr(f), nIn, V, ptr(sp_got), nOut, V, ptr(ga.blocks_wide(Ts)), Ts, vol,
                                                    None, 0, ptr(pf), P, ptr(scratch), ptr(res), stream()))
    assert torch.equal(sp_got, sp_plain + res)
    # ---- BatchNorm backward with the gradient sum of a second consumer (planes = nOut; large and small-map kernels)
    for rows in (V, min(V, 700)):
        xb, dy, add = f[:rows, :min(nIn, nOut)].contiguous(), None, None
        planes = xb.shape[1]
        xb = torch.as_tensor(rng.standard_normal((rows, planes)).astype(np.float32)).to(DEV).bfloat16()
        dy = torch.as_tensor(rng.standard_normal((rows, planes)).astype(np.float32)).to(DEV).bfloat16()
        add = torch.as_tensor(rng.standard_normal((rows, planes)).astype(np.float32)).to(DEV).bfloat16()
        w_ = torch.rand(planes, device=DEV) + 0.5
        b_ = torch.randn(planes, device=DEV)
        rm, rv = torch.zeros(planes, device=DEV), torch.ones(planes, device=DEV)
        mean, inv = torch.empty(planes, device=DEV), torch.empty(planes, device=DEV)
        yb = torch.empty_like(xb)
        ws = torch.empty(int(lib.aabr_bn_scratch_floats(planes)), device=DEV)
        check(lib.aabr_bn_forward_bf16(ptr(xb), ptr(yb), rows, planes, ptr(mean), ptr(inv), ptr(rm), ptr(rv), ptr(w_), ptr(b_),
                                       1e-4, 0.9, 1, 0.2, ptr(ws), stream()))
        dx0, dw0, db0 = torch.empty_like(xb), torch.empty(planes, device=DEV), torch.empty(planes, device=DEV)
        check(lib.aabr_bn_backward_bf16(ptr(xb), ptr(dx0), ptr(yb), ptr(dy), rows, planes, ptr(mean), ptr(inv), ptr(w_), ptr(b_),
                                        ptr(dw0), ptr(db0), 0.2, ptr(ws), stream()))
        dx1, dw1, db1 = torch.empty_like(xb), torch.empty(planes, device=DEV), torch.empty(planes, device=DEV)
        check(lib.aabr_bn_backward_add_bf16(ptr(xb), ptr(dx1), ptr(yb), ptr(dy), rows, planes, ptr(mean), ptr(inv), ptr(w_),
                                            ptr(b_), ptr(dw1), ptr(db1), 0.2, None, 0, ptr(ws), ptr(add), stream()))
        assert torch.equal(dx1, dx0 + add) and torch.equal(dw1, dw0) and torch.equal(db1, db0)

"""k_conv_narrow (csrc/conv_narrow.hip): the 32 -> 32 plane layers read from the gather table, all offsets' weights in
LDS, 16 output rows per wave in registers -- against the oracle (SCN/CPU/Convolution.cpp:46-79,117-185;
Deconvolution.cpp:7-77) through the C ABI and through the layers (submanifold, strided, transposed; forward and
backward), fp32 and bf16 storage."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scn():
    importlib.import_module("automatic-as-built-reconstruction_amd")
    import sparseconvnet as scn
    return scn


def _t(a):
    return torch.as_tensor(a).to(DEV)


@pytest.fixture
def force_narrow():
    import _hip
    _hip.set_knob("CONV_NARROW", 1)
    yield
    _hip.set_knob("CONV_NARROW", None)


def _scene(rng, n, size, batch, C):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64), rng.standard_normal((n, C)).astype(np.float32)


def _match_rows(dev_coords, ref_coords):
    """oracle row of every device row (the same sites, matched by their coordinates)"""
    key = lambda c: ((np.asarray(c[:, 3], np.int64) * 70000 + c[:, 0]) * 70000 + c[:, 1]) * 70000 + c[:, 2]
    kd, kr = key(dev_coords), key(ref_coords)
    o = np.argsort(kr)
    pos = np.searchsorted(kr[o], kd)
    assert (kr[o][pos] == kd).all()
    return o[pos]


def _variant():
    import _hip
    return _hip.load().aabr_conv_last_variant().decode()


@pytest.mark.parametrize("npts,bf", [(700, False), (3001, False), (2500, True), (17, True), (1, False)])
def test_narrow_c_abi_forward_and_input_gradient_forms(npts, bf):
    """aabr_conv_forward_narrow[_bf16]: forward (with bias) and the submanifold input-gradient form (transposed weights,
    mirrored offsets) on ragged row counts (V % 16 != 0, V < 16); same call twice, same bits"""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(npts)
    coords, _ = _scene(rng, npts, (12, 11, 5), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = (rng.standard_normal((vol, 1, 32, 32)) * 0.1).astype(np.float32)
    Wd = _t(W)
    dt = torch.bfloat16 if bf else torch.float32
    fn = lib.aabr_conv_forward_narrow_bf16 if bf else lib.aabr_conv_forward_narrow
    Wr = (Wd.bfloat16().float() if bf else Wd).cpu().numpy().reshape(vol, 32, 32)
    f = _t(rng.standard_normal((V, 32)).astype(np.float32)).to(dt)
    b = rng.standard_normal(32).astype(np.float32)
    tol = dict(rtol=2 ** -7, atol=2 ** -7) if bf else dict(rtol=1e-4, atol=2e-5)
    out = torch.full((V, 32), float("nan"), dtype=dt, device=DEV)
    check(fn(ptr(f), V, ptr(out), V, ptr(ga.table), vol, ptr(Wd), ptr(_t(b)), 0, stream()))
    assert _variant().startswith("k_conv_narrow<")
    ref, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, V, b)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=tol["rtol"], atol=tol["atol"] * max(np.abs(ref).max(), 1.0))
    out2 = torch.empty_like(out)
    check(fn(ptr(f), V, ptr(out2), V, ptr(ga.table), vol, ptr(Wd), ptr(_t(b)), 0, stream()))
    assert torch.equal(out, out2)
    g = _t(rng.standard_normal((V, 32)).astype(np.float32)).to(dt)
    d_in = torch.full((V, 32), float("nan"), dtype=dt, device=DEV)
    check(fn(ptr(g), V, ptr(d_in), V, ptr(ga.table), vol, ptr(Wd), None, 3, stream()))
    dref, _, _ = O.conv_bwd(np.zeros((V, 32), np.float32), g.float().cpu().numpy(), Wr, rb, want_bias=False)
    np.testing.assert_allclose(d_in.float().cpu().numpy(), dref, rtol=tol["rtol"], atol=tol["atol"] * max(np.abs(dref).max(), 1.0))
    # argument checks
    assert fn(ptr(f), V, ptr(out), V, ptr(ga.table), 29, ptr(Wd), None, 0, stream()) != 0
    assert fn(ptr(f), V, ptr(out), V, None, vol, ptr(Wd), None, 0, stream()) != 0
    assert lib.aabr_conv_narrow_ok(32, 64, V, V, vol, 1) == 0 and lib.aabr_conv_narrow_ok(64, 32, V, V, vol, 1) == 0


@pytest.mark.parametrize("bf,order", [(True, "first_seen"), (True, "brick"), (False, "brick")])
def test_narrow_c_abi_at_its_dispatch_size(bf, order):
    """aabr_conv_forward_narrow[_bf16] through the C ABI against the oracle at >= 400 k output rows --
    the size from which the library dispatches it by default (aabr_conv_narrow_ok without a knob) -- on a scene-shaped
    rule book (1.5 M-point scene @ 2 cm), forward with bias and the mirrored input-gradient form; rows matched by
    coordinates when the grid is brick-ordered"""
    import _hip
    import synth_scenes as S
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(77)
    l, _ = S.make_scene(750000, 31, 50)
    coords = np.concatenate([l, np.zeros((l.shape[0], 1), np.int64)], 1)
    layer = scn.InputLayer(3, [4096, 4096, 512], mode=4)
    layer.site_order = order
    x = layer([_t(coords), _t(np.zeros((coords.shape[0], 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    assert V >= 400000 and lib.aabr_conv_narrow_ok(32, 32, V, V, vol, 1) == 1       # default dispatch: no knob set
    O.set_threads(16)
    il = O.input_layer(coords, np.zeros((coords.shape[0], 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    assert il["V"] == V
    if order == "brick":
        r = _match_rows(x.get_spatial_locations().numpy(), il["coords"])         # device row i = oracle row r[i]
    else:
        np.testing.assert_array_equal(x.get_spatial_locations().numpy(), il["coords"])
        r = np.arange(V)
    W = (rng.standard_normal((vol, 1, 32, 32)) * 0.1).astype(np.float32)
    Wd = _t(W)
    dt = torch.bfloat16 if bf else torch.float32
    fn = lib.aabr_conv_forward_narrow_bf16 if bf else lib.aabr_conv_forward_narrow
    Wr = (Wd.bfloat16().float() if bf else Wd).cpu().numpy().reshape(vol, 32, 32)
    f_o = rng.standard_normal((V, 32)).astype(np.float32)       # oracle row order
    f = _t(f_o[r]).to(dt)
    b = rng.standard_normal(32).astype(np.float32)
    tol = dict(rtol=2 ** -7, atol=2 ** -7) if bf else dict(rtol=1e-4, atol=2e-5)
    out = torch.full((V, 32), float("nan"), dtype=dt, device=DEV)
    check(fn(ptr(f), V, ptr(out), V, ptr(ga.table), vol, ptr(Wd), ptr(_t(b)), 0, stream()))
    assert _variant().startswith("k_conv_narrow<")
    f_used = np.empty_like(f_o)
    f_used[r] = f.float().cpu().numpy()                          # what the device read (bf16-rounded), oracle order
    ref, _ = O.conv_fwd(f_used, Wr, rb, V, b)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref[r], rtol=tol["rtol"], atol=tol["atol"] * max(np.abs(ref).max(), 1.0))
    g_o = rng.standard_normal((V, 32)).astype(np.float32)
    g = _t(g_o[r]).to(dt)
    d_in = torch.full((V, 32), float("nan"), dtype=dt, device=DEV)
    check(fn(ptr(g), V, ptr(d_in), V, ptr(ga.table), vol, ptr(Wd), None, 3, stream()))
    g_used = np.empty_like(g_o)
    g_used[r] = g.float().cpu().numpy()
    dref, _, _ = O.conv_bwd(np.zeros((V, 32), np.float32), g_used, Wr, rb, want_bias=False)
    np.testing.assert_allclose(d_in.float().cpu().numpy(), dref[r], rtol=tol["rtol"], atol=tol["atol"] * max(np.abs(dref).max(), 1.0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_narrow_through_the_layers(force_narrow, dtype):
    """SubmanifoldConvolution / Convolution / Deconvolution with 32 -> 32 planes dispatch the narrow kernel for their
    forward and input-gradient launches (strided books: the output-side table forward, the input-side table backward;
    transposed: the other way round); outputs, input gradient and weight gradients against the oracle"""
    scn = _scn()
    bf = dtype == torch.bfloat16
    rng = np.random.default_rng(5)
    size = np.array([16, 16, 8])
    coords, feats = _scene(rng, 2500, tuple(size), 2, 32)
    f = _t(feats).requires_grad_(True)
    x = scn.InputLayer(3, list(size), mode=4)([_t(coords), f])
    sub = scn.SubmanifoldConvolution(3, 32, 32, 3, True).to(DEV)
    conv = scn.Convolution(3, 32, 32, [2, 2, 2], [2, 2, 2], False).to(DEV)
    dec = scn.Deconvolution(3, 32, 32, [2, 2, 2], [2, 2, 2], False).to(DEV)
    xs = x
    if bf:
        xs = scn.SparseConvNetTensor()
        xs.metadata, xs.spatial_size = x.metadata, x.spatial_size
        xs.features = x.features.to(torch.bfloat16)
    y0 = sub(xs)
    assert _variant().startswith("k_conv_narrow<"), _variant()
    y1 = conv(y0)
    assert _variant().startswith("k_conv_narrow<"), _variant()
    y2 = dec(y1)
    assert _variant().startswith("k_conv_narrow<"), _variant()
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    osz = (size - 2) // 2 + 1
    rbs, oc = O.convolution_rules(il["coords"], [2, 2, 2], [2, 2, 2], osz)
    rd = (lambda a: torch.as_tensor(a).bfloat16().float().numpy()) if bf else (lambda a: a)
    Ws = rd(sub.weight.detach().cpu().numpy().reshape(27, 32, 32))
    Wc = rd(conv.weight.detach().cpu().numpy().reshape(8, 32, 32))
    Wd = rd(dec.weight.detach().cpu().numpy().reshape(8, 32, 32))
    r0, _ = O.conv_fwd(rd(il["out"]), Ws, rb, il["V"], sub.bias.detach().cpu().numpy())
    r0 = rd(r0)
    r1, _ = O.conv_fwd(r0, Wc, rbs, oc.shape[0])
    r1 = rd(r1)
    r2, _ = O.conv_fwd(r1, Wd, rbs, il["V"], in_col=1)
    tol = 2 ** -6 if bf else 1e-4
    for got, ref in ((y0, r0), (y1, r1), (y2, r2)):
        np.testing.assert_allclose(got.features.detach().float().cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())
    g = rng.standard_normal(r2.shape).astype(np.float32)
    y2.features.backward(_t(g).to(dtype))
    assert f.grad is not None
    d1, dWd, _ = O.conv_bwd(r1, rd(g), Wd, rbs, in_col=1)
    d0, dWc, _ = O.conv_bwd(r0, rd(d1), Wc, rbs)
    dx, dWs, _ = O.conv_bwd(rd(il["out"]), rd(d0), Ws, rb)
    d_feats = O.input_layer_bwd(il, dx)
    gt = 4 * tol
    np.testing.assert_allclose(f.grad.cpu().numpy(), d_feats, rtol=gt, atol=gt * np.abs(d_feats).max())
    np.testing.assert_allclose(sub.weight.grad.cpu().numpy().reshape(Ws.shape), dWs, rtol=gt, atol=gt * np.abs(dWs).max())
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(Wc.shape), dWc, rtol=gt, atol=gt * np.abs(dWc).max())
    np.testing.assert_allclose(dec.weight.grad.cpu().numpy().reshape(Wd.shape), dWd, rtol=gt, atol=gt * np.abs(dWd).max())


def test_narrow_default_dispatch_threshold():
    """without the knob the narrow kernel takes bf16-storage launches of >= 400,000 output rows only (smaller ones and
    fp32 storage keep the tile kernels: measured equal / slower there), and the choice is a pure function of the shape"""
    import _hip
    _scn()
    lib = _hip.load()
    assert lib.aabr_conv_narrow_ok(32, 32, 500000, 400000, 27, 1) == 1
    assert lib.aabr_conv_narrow_ok(32, 32, 500000, 399999, 27, 1) == 0
    assert lib.aabr_conv_narrow_ok(32, 32, 500000, 400000, 27, 0) == 0
    assert lib.aabr_conv_narrow_ok(32, 32, 500000, 400000, 29, 1) == 0
    assert lib.aabr_conv_narrow_ok(32, 32, 0, 400000, 27, 1) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_narrow_equals_tile_kernel_at_scene_size(dtype):
    """two 80 k-point scenes at 2 cm (~150 k active rows: every workgroup of the persistent grid sweeps several row
    groups, the last group is ragged): the narrow kernel against the 64-row-tile kernel on the same submanifold book --
    fp32: same sums in another order (1e-5); bf16: both round once at the store (one bf16 step)"""
    import _hip
    import synth_scenes as S
    scn = _scn()
    locs, _ = S.make_batch(2, 80000, 77, 50)
    n = locs.shape[0]
    x = scn.InputLayer(3, [4096, 4096, 512], mode=4)([_t(locs), _t(np.zeros((n, 1), np.float32))])
    V = x.features.shape[0]
    assert V > 100000
    rng = np.random.default_rng(3)
    conv = scn.SubmanifoldConvolution(3, 32, 32, 3, False).to(DEV)
    xs = scn.SparseConvNetTensor()
    xs.metadata, xs.spatial_size = x.metadata, x.spatial_size
    xs.features = _t(rng.standard_normal((V, 32)).astype(np.float32)).to(dtype)
    outs = []
    for knob in (1, 0):
        _hip.set_knob("CONV_NARROW", knob)
        try:
            with torch.no_grad():
                y = conv(xs)
            outs.append((y.features.float().clone(), _variant()))
        finally:
            _hip.set_knob("CONV_NARROW", None)
    assert outs[0][1].startswith("k_conv_narrow<") and not outs[1][1].startswith("k_conv_narrow<")
    a, b = outs[0][0], outs[1][0]
    tol = 2 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    assert float(b.abs().max()) > 0.1


def test_narrow_write_out_statistics_feed_batchnorm():
    """aabr_conv_forward_narrow_bf16_stats / _bwd_stats: the per-workgroup fp64 partial sums add up to numpy's sums over the
    STORED bf16 values (forward: v, v^2; backward: masked d and (x - mean) d with the sign from the BatchNorm's stored
    output, SCN/CPU/BatchNormalization.cpp:28-40,66-84 on the bf16-storage model), the outputs are the bits of the plain
    launch, and aabr_bn_forward_parts_bf16 / aabr_bn_backward_parts_bf16 fed with them agree with the BatchNorm's own
    statistics passes."""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(12)
    npts = 5000
    coords, _ = _scene(rng, npts, (14, 13, 6), 2, 1)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    nparts = int(lib.aabr_conv_narrow_parts(V))
    assert nparts == min(256, ((V + 15) // 16 + 15) // 16) and nparts > 1
    W = _t((rng.standard_normal((vol, 1, 32, 32)) * 0.1).astype(np.float32))
    f = _t(rng.standard_normal((V, 32)).astype(np.float32)).bfloat16()
    leak = 0.2
    _hip.set_knob("BN_SMALL", 0)
    try:
        out0 = torch.empty((V, 32), dtype=torch.bfloat16, device=DEV)
        check(lib.aabr_conv_forward_narrow_bf16(ptr(f), V, ptr(out0), V, ptr(ga.table), vol, ptr(W), None, 0, stream()))
        # forward statistics
        out = torch.empty_like(out0)
        st = torch.full((nparts, 2, 32), float("nan"), dtype=torch.float64, device=DEV)
        check(lib.aabr_conv_forward_narrow_bf16_stats(ptr(f), V, ptr(out), V, ptr(ga.table), vol, ptr(W), None, 0, ptr(st),
                                                      stream()))
        assert torch.equal(out, out0)
        o = out.float().cpu().numpy().astype(np.float64)
        s = st.cpu().numpy().sum(0)
        np.testing.assert_allclose(s[0], o.sum(0), rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(s[1], (o ** 2).sum(0), rtol=1e-12, atol=1e-10)
        ws = torch.empty(int(lib.aabr_bn_scratch_floats(32)), device=DEV)
        gam = _t(rng.uniform(0.5, 1.5, 32).astype(np.float32))
        bet = _t(rng.standard_normal(32).astype(np.float32))

        def bn_fwd(parts):
            y = torch.empty_like(out)
            sm, si, rm, rv = (torch.zeros(32, device=DEV) for _ in range(4))
            a = (ptr(out), ptr(y), V, 32, ptr(sm), ptr(si), ptr(rm), ptr(rv), ptr(gam), ptr(bet), 1e-4, 0.9)
            if parts:
                check(lib.aabr_bn_forward_parts_bf16(*a, leak, ptr(st), nparts, ptr(ws), stream()))
            else:
                check(lib.aabr_bn_forward_bf16(*a, 1, leak, ptr(ws), stream()))
            return y, sm, si

        (y1, sm1, si1), (y0, sm0, si0) = bn_fwd(True), bn_fwd(False)
        np.testing.assert_allclose(sm1.cpu().numpy(), sm0.cpu().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(si1.cpu().numpy(), si0.cpu().numpy(), rtol=1e-6)
        assert float((y1.float() - y0.float()).abs().max()) <= 2 ** -7 * float(y0.float().abs().max())
        # backward statistics: this launch writes the d_out of the BatchNorm (input xb, stored output yb)
        xb = _t((rng.standard_normal((V, 32)) * 1.3 + 0.2).astype(np.float32)).bfloat16()
        yb = torch.empty_like(xb)
        sm, si, rm, rv = (torch.zeros(32, device=DEV) for _ in range(4))
        check(lib.aabr_bn_forward_bf16(ptr(xb), ptr(yb), V, 32, ptr(sm), ptr(si), ptr(rm), ptr(rv), ptr(gam), ptr(bet), 1e-4,
                                       0.9, 1, leak, ptr(ws), stream()))
        d_out = torch.empty_like(out0)
        stb = torch.full((nparts, 2, 32), float("nan"), dtype=torch.float64, device=DEV)
        check(lib.aabr_conv_forward_narrow_bf16_bwd_stats(ptr(f), V, ptr(d_out), V, ptr(ga.table), vol, ptr(W), None, 0,
                                                          ptr(stb), ptr(xb), ptr(yb), ptr(sm), leak, stream()))
        assert torch.equal(d_out, out0)
        d32 = d_out.float().cpu().numpy()
        dm = np.where(yb.float().cpu().numpy() > 0, d32, d32 * np.float32(leak)).astype(np.float64)
        xc = (xb.float().cpu().numpy() - sm.cpu().numpy().astype(np.float32)).astype(np.float64)
        sb = stb.cpu().numpy().sum(0)
        np.testing.assert_allclose(sb[0], dm.sum(0), rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(sb[1], (xc * dm).sum(0), rtol=1e-12, atol=1e-10)

        def bn_bwd(parts):
            d_in = torch.empty_like(xb)
            dw, db = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
            common = (ptr(xb), ptr(d_in), ptr(yb), ptr(d_out), V, 32, ptr(sm), ptr(si), ptr(gam), ptr(bet), ptr(dw), ptr(db),
                      leak)
            if parts:
                check(lib.aabr_bn_backward_parts_bf16(*common, ptr(stb), nparts, ptr(ws), stream()))
            else:
                check(lib.aabr_bn_backward_bf16(*common, ptr(ws), stream()))
            return d_in.float().cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()

        got, want = bn_bwd(True), bn_bwd(False)
        np.testing.assert_allclose(got[2], want[2], rtol=3e-7, atol=1e-6)
        np.testing.assert_allclose(got[1], want[1], rtol=3e-7, atol=1e-6)
        diff = np.abs(got[0] - want[0])
        assert (diff > 0).mean() < 1e-3 and np.all(diff <= 2 ** -7 * np.abs(want[0]) + 1e-30)
    finally:
        _hip.set_knob("BN_SMALL", None)


def test_narrow_in_the_compiled_graph_with_fused_statistics(force_narrow):
    """FPN_Net in bf16 storage with the narrow kernel forced for its 32 -> 32 layers: the compiled graph (narrow records,
    BatchNorm statistics from their write-outs) equals the module path (narrow launches, BatchNorm's own statistics) in
    outputs and input gradient to the bf16 tolerance, and parameter gradients agree"""
    import synth_scenes as S
    from test_gpu_fpn import _fpn
    scn = _scn()
    from sparseconvnet import planExecutor
    keep = planExecutor.narrow_stats
    planExecutor.narrow_stats = True          # (off by default: measured a loss on the step; the wiring stays tested)
    try:
        _narrow_compiled_graph_body(scn, S, _fpn)
    finally:
        planExecutor.narrow_stats = keep


def _narrow_compiled_graph_body(scn, S, _fpn):
    torch.manual_seed(11)
    net = _fpn(feature_dtype=torch.bfloat16).to(DEV)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    locs, feats = S.make_batch(2, 20000, 47, 20)
    l = _t(locs)

    def run(compiled):
        net.load_state_dict(state)
        net.train(True)
        net.compiled_graph = compiled
        f = _t(feats).requires_grad_(True)
        net.zero_grad()
        rpn, roi = net([l, f])
        outs = [m.features.detach().float().clone() for m in rpn + roi]
        sum(m.features.float().square().mean() for m in rpn + roi).backward()
        torch.cuda.synchronize()
        return outs, f.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    a, b = run(False), run(True)
    for x, y in zip(a[0], b[0]):
        assert float((x - y).abs().max()) <= 2 ** -5 * float(x.abs().max())
    assert float((a[1] - b[1]).abs().max()) <= 2e-2 * float(a[1].abs().max())
    for n in a[2]:
        ga, gb = a[2][n].float(), b[2][n].float()
        assert float((ga - gb).abs().max()) <= 3e-2 * float(ga.abs().max()) + 1e-12, n

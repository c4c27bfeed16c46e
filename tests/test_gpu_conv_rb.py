"""k_conv_rb (csrc/conv_rb.hip): wide layers in bf16 storage from the gather table with register accumulators over
256-row tiles -- through the C ABI against the oracle (SCN/CPU/Convolution.cpp:46-79,117-185) on bf16-rounded operands:
forward, the mirrored submanifold input-gradient form, the strided forms (output-side and input-side tables, transposed
pack), ragged row counts (V % 256 != 0, V < 16), all four plane shapes, both site orders; same call twice, same bits."""
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
    return torch.as_tensor(a).to(DEV)


def _scene(rng, n, size, batch):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64)


def _match_rows(dev_coords, ref_coords):
    """oracle row of every device row (the same sites, matched by their coordinates)"""
    key = lambda c: ((np.asarray(c[:, 3], np.int64) * 70000 + c[:, 0]) * 70000 + c[:, 1]) * 70000 + c[:, 2]
    kd, kr = key(dev_coords), key(ref_coords)
    o = np.argsort(kr)
    pos = np.searchsorted(kr[o], kd)
    assert (kr[o][pos] == kd).all()
    return o[pos]


def _pack(lib, Wd, vol, a, b):
    from _hip import ptr, stream, check
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, a, b))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, a, b, ptr(pf), ptr(pt), stream()))
    return pf, pt


@pytest.mark.parametrize("nIn,nOut,npts,order", [(128, 128, 3000, "brick"), (64, 64, 2500, "first_seen"), (64, 128, 700, "brick"),
                                                 (128, 64, 9, "brick"), (128, 128, 1, "first_seen")])
def test_rb_submanifold_forward_and_input_gradient(nIn, nOut, npts, order):
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + nOut + npts)
    coords = _scene(rng, npts, (24, 19, 6), 2)
    layer = scn.InputLayer(3, [32, 32, 8], mode=4)
    layer.site_order = order
    x = layer([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    r = _match_rows(x.get_spatial_locations().numpy(), il["coords"])    # device row i = oracle row r[i]
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    pf, pt = _pack(lib, Wd, vol, nIn, nOut)
    assert lib.aabr_conv_rb_ok(nIn, nOut, V, V, vol) == 0          # not dispatched by default ...
    _hip.set_knob("CONV_RB", 1)
    try:
        assert lib.aabr_conv_rb_ok(nIn, nOut, V, V, vol) == 1      # ... the knob turns it on
        assert lib.aabr_conv_rb_ok(256, 128, V, V, vol) == 0 and lib.aabr_conv_rb_ok(nIn, nOut, V, V, 28) == 0
    finally:
        _hip.set_knob("CONV_RB", None)
    f_o = rng.standard_normal((V, nIn)).astype(np.float32)
    f = _t(f_o[r]).bfloat16()
    f_used = np.empty_like(f_o)
    f_used[r] = f.float().cpu().numpy()
    b = rng.standard_normal(nOut).astype(np.float32)
    out = torch.full((V, nOut), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rb_bf16(ptr(f), nIn, V, ptr(out), nOut, V, ptr(ga.table), vol, ptr(_t(b)), 0, ptr(pf), stream()))
    assert lib.aabr_conv_last_variant().decode().startswith("k_conv_rb<")
    ref, _ = O.conv_fwd(f_used, Wr, rb, V, b)
    tol = 2.0 ** -7
    np.testing.assert_allclose(out.float().cpu().numpy(), ref[r], rtol=tol, atol=tol * max(np.abs(ref).max(), 1.0))
    out2 = torch.empty_like(out)
    check(lib.aabr_conv_forward_rb_bf16(ptr(f), nIn, V, ptr(out2), nOut, V, ptr(ga.table), vol, ptr(_t(b)), 0, ptr(pf), stream()))
    assert torch.equal(out, out2)
    # input gradient of the same layer: d_out [V, nOut] -> d_in [V, nIn], transposed pack, mirrored offsets
    g_o = rng.standard_normal((V, nOut)).astype(np.float32)
    g = _t(g_o[r]).bfloat16()
    g_used = np.empty_like(g_o)
    g_used[r] = g.float().cpu().numpy()
    d_in = torch.full((V, nIn), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rb_bf16(ptr(g), nOut, V, ptr(d_in), nIn, V, ptr(ga.table), vol, None, 3, ptr(pt), stream()))
    dref, _, _ = O.conv_bwd(np.zeros((V, nIn), np.float32), g_used, Wr, rb, want_bias=False)
    np.testing.assert_allclose(d_in.float().cpu().numpy(), dref[r], rtol=tol, atol=tol * max(np.abs(dref).max(), 1.0))
    # argument checks
    assert lib.aabr_conv_forward_rb_bf16(ptr(f), 96, V, ptr(out), nOut, V, ptr(ga.table), vol, None, 0, ptr(pf), stream()) != 0
    assert lib.aabr_conv_forward_rb_bf16(ptr(f), nIn, V, ptr(out), nOut, V, None, vol, None, 0, ptr(pf), stream()) != 0


@pytest.mark.parametrize("nIn,nOut", [(64, 128), (128, 128)])
def test_rb_strided_forms(nIn, nOut):
    """Convolution forward (output-side table), its input gradient (input-side table, transposed pack) and the
    Deconvolution forward over the same book (input-side table), 2/2 filter"""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(5 + nIn)
    coords = _scene(rng, 4000, (32, 32, 8), 2)
    layer = scn.InputLayer(3, [32, 32, 8], mode=4)
    layer.site_order = "brick"
    x = layer([_t(coords), _t(np.zeros((4000, 1), np.float32))])
    tb = x.metadata.getRuleBook(x.spatial_size, torch.LongTensor([16, 16, 4]), torch.LongTensor([2, 2, 2]),
                                torch.LongTensor([2, 2, 2]))
    il = O.input_layer(coords, np.zeros((4000, 1), np.float32), 4)
    rb, oc = O.convolution_rules(il["coords"], [2, 2, 2], [2, 2, 2], [16, 16, 4])
    ri = _match_rows(x.get_spatial_locations().numpy(), il["coords"])
    loc = x.metadata.getSpatialLocations(torch.LongTensor([16, 16, 4])).numpy()
    key = lambda c: ((c[:, 3] * 64 + c[:, 0]) * 64 + c[:, 1]) * 64 + c[:, 2]
    o = np.argsort(key(oc))
    ro = o[np.searchsorted(key(oc)[o], key(loc))]                     # device output row -> oracle output row
    V_in, V_out, vol = tb.V_in, tb.V_out, tb.vol
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    pf, pt = _pack(lib, Wd, vol, nIn, nOut)
    tol = 2.0 ** -7
    f_o = rng.standard_normal((V_in, nIn)).astype(np.float32)
    f = _t(f_o[ri]).bfloat16()
    f_used = np.empty_like(f_o)
    f_used[ri] = f.float().cpu().numpy()
    out = torch.full((V_out, nOut), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rb_bf16(ptr(f), nIn, V_in, ptr(out), nOut, V_out, ptr(tb.out.table), vol, None, 0, ptr(pf), stream()))
    ref, _ = O.conv_fwd(f_used, Wr, rb, V_out)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref[ro], rtol=tol, atol=tol * np.abs(ref).max())
    g_o = rng.standard_normal((V_out, nOut)).astype(np.float32)
    g = _t(g_o[ro]).bfloat16()
    g_used = np.empty_like(g_o)
    g_used[ro] = g.float().cpu().numpy()
    d_in = torch.full((V_in, nIn), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rb_bf16(ptr(g), nOut, V_out, ptr(d_in), nIn, V_in, ptr(tb.inn.table), vol, None, 1, ptr(pt), stream()))
    dref, _, _ = O.conv_bwd(np.zeros((V_in, nIn), np.float32), g_used, Wr, rb, want_bias=False)
    np.testing.assert_allclose(d_in.float().cpu().numpy(), dref[ri], rtol=tol, atol=tol * np.abs(dref).max())

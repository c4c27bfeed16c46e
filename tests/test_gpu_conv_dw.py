"""Weight gradient of the sparse convolutions (reference: SCN/CPU/Convolution.cpp:81-114, `dW[k] += X[rules_in]^T dY[rules_out]`)
through the round-5 kernels: the full-tile form (`k_conv_dw_full_f32 / _bf16`: one workgroup per chunk forms a whole
128 x 128 block of dW, rows gathered once and shared through LDS) and the direct form (every offset at most one chunk:
no partial buffer, no reduce launch), against the oracle, against the 64 x 64-block kernels (knob DW_FULL = 0) and for
bit-reproducibility."""
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


def _bf(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _rand_scene(rng, n, size, batch, C):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64), rng.standard_normal((n, C)).astype(np.float32)


def _dw_once(scn, x_md, spatial, feats_t, g_t, nIn, nOut, bias):
    """dW (and d_bias) of a 3x3x3 submanifold layer for the given site features and output gradient"""
    import _hip
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, bias).to(DEV)
    leaf = feats_t.clone()           # no input gradient asked for: the weight gradient is the last launch
    x = scn.SparseConvNetTensor()
    x.metadata, x.spatial_size, x.features = x_md, spatial, leaf
    y = conv(x)
    with torch.autograd.set_multithreading_enabled(False):      # aabr_conv_last_variant() is per thread
        y.features.backward(g_t)
    lib = _hip.load()
    return (conv.weight.grad.detach().clone(), conv.bias.grad.detach().clone() if bias else None,
            lib.aabr_conv_last_variant().decode())


@pytest.fixture(autouse=True)
def _full_tile_from_small_rule_books():
    """the library hands rule books to the full-tile kernels from ~128 workgroups on; the tests' small grids reach them
    with the threshold lowered"""
    import _hip
    _hip.set_knob("DW_FULL_MIN", 8)
    yield
    _hip.set_knob("DW_FULL_MIN", None)


@pytest.mark.parametrize("bf", [False, True])
@pytest.mark.parametrize("nIn,nOut,npts,size,full", [
    (128, 128, 150, (6, 6, 4), False),         # V <= chunk: the direct form (one launch, no partial buffer)
    (256, 256, 700, (10, 10, 5), False),       # direct form, sixteen 64 x 64 blocks
    (128, 128, 2500, (14, 12, 6), True),       # full-tile kernel, 48 workgroups: ranges cross offset boundaries
    (128, 256, 2500, (14, 12, 6), True),
    (256, 128, 2500, (14, 12, 6), True),
    (128, 128, 30000, (40, 36, 12), True),     # several workgroups per offset: the reduce's workgroup order
    (256, 256, 12000, (30, 30, 10), True),     # four 128 x 128 blocks
    (64, 64, 200, (6, 6, 4), False),           # direct form of the 64 x 64-block kernels (256-pair chunks)
    (32, 64, 900, (10, 10, 5), False),         # 1024-pair chunks, direct
    (64, 64, 6000, (20, 20, 8), False),        # chunked form + reduce (unchanged path)
])
def test_weight_gradient_full_tile_and_direct_forms(bf, nIn, nOut, npts, size, full):
    import _hip
    scn = _scn()
    rng = np.random.default_rng(nIn * 31 + nOut + npts)
    coords, feats = _rand_scene(rng, npts, size, 2, nIn)
    spatial = [s + 2 for s in size]
    x = scn.InputLayer(3, spatial, mode=4)([_t(coords), _t(feats)])
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    xin = _bf(il["out"]) if bf else il["out"]
    g = rng.standard_normal((il["V"], nOut)).astype(np.float32)
    g = _bf(g) if bf else g
    dt = torch.bfloat16 if bf else torch.float32
    f_t, g_t = _t(xin).to(dt), _t(g).to(dt)
    W0 = np.zeros((27, nIn, nOut), np.float32)
    _, dW_ref, db_ref = O.conv_bwd(xin, g, W0, rb, want_bias=True)

    dW, db, variant = _dw_once(scn, x.metadata, x.spatial_size, f_t, g_t, nIn, nOut, True)
    assert ("k_conv_dw_full" in variant) == full, variant
    got = dW.cpu().numpy().reshape(27, nIn, nOut)
    np.testing.assert_allclose(got, dW_ref, rtol=1e-4, atol=1e-5 * np.abs(dW_ref).max())
    np.testing.assert_allclose(db.cpu().numpy(), db_ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(db_ref).max()))
    # offsets without a single rule (possible on the tiny grids) must come out as exact zeros in the direct form
    for k in range(27):
        if int(rb.counts[k]) == 0:
            assert not got[k].any()
    # bit-reproducible
    dW2, _, _ = _dw_once(scn, x.metadata, x.spatial_size, f_t, g_t, nIn, nOut, True)
    assert torch.equal(dW, dW2)
    # the 64 x 64-block kernels on the same operands: the same sums in another order
    if full:
        _hip.set_knob("DW_FULL", 0)
        try:
            dW3, _, v3 = _dw_once(scn, x.metadata, x.spatial_size, f_t, g_t, nIn, nOut, True)
        finally:
            _hip.set_knob("DW_FULL", None)
        assert "k_conv_dw_pairs" in v3
        np.testing.assert_allclose(dW3.cpu().numpy().reshape(27, nIn, nOut), got, rtol=2e-5,
                                   atol=2e-6 * np.abs(dW_ref).max())


def test_direct_form_issues_no_reduce_launch():
    """V <= chunk_pairs: ONE kernel forms dW (the scratch buffer is never written: poisoned before the call, NaNs stay)"""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(5)
    coords, feats = _rand_scene(rng, 300, (8, 8, 4), 1, 128)
    x = scn.InputLayer(3, [10, 10, 6], mode=4)([_t(coords), _t(feats)])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    V, vol = tb.V_out, tb.vol
    assert V <= lib.aabr_conv_dw_chunk_pairs(V, vol, 128, 128)
    pairs = tb.out.pairs()
    mc = tb.out.max_chunks(128, 128)
    scn.SCN.flush_geom()
    g = torch.randn(V, 128, device=DEV)
    scratch = torch.full((int(lib.aabr_conv_dw_scratch_floats(mc, 128, 128)),), float("nan"), device=DEV)
    dW = torch.empty(vol, 128, 128, device=DEV)
    check(lib.aabr_conv_backward_weight(ptr(x.features), 128, ptr(g), 128, V, ptr(pairs), vol, mc, ptr(dW), None,
                                        ptr(scratch), stream()))
    torch.cuda.synchronize()
    assert torch.isnan(scratch).all() and torch.isfinite(dW).all()
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    _, dW_ref, _ = O.conv_bwd(il["out"], g.cpu().numpy(), np.zeros((27, 128, 128), np.float32), rb, want_bias=False)
    np.testing.assert_allclose(dW.cpu().numpy(), dW_ref, rtol=1e-4, atol=1e-5 * np.abs(dW_ref).max())


@pytest.mark.parametrize("bf", [False, True])
def test_default_dispatch_by_rule_book_size(bf):
    """without knobs: a scene-sized 128 -> 128 rule book (77 k rows, 3^3) goes to the full-tile kernel with a multiple of
    256 workgroups, its 1x1x1 rule book and a small grid stay with the 64 x 64-block kernels; both forms agree"""
    import _hip
    import synth_scenes as S
    scn = _scn()
    _hip.set_knob("DW_FULL_MIN", None)
    l, _ = S.make_batch(1, 80000, 9000, 50)
    x = scn.InputLayer(3, [4096, 4096, 512], mode=4)([_t(l), _t(np.zeros((l.shape[0], 1), np.float32))])
    V = x.features.size(0)
    dt = torch.bfloat16 if bf else torch.float32
    torch.manual_seed(3)
    f_t, g_t = torch.randn(V, 128, device=DEV).to(dt), torch.randn(V, 128, device=DEV).to(dt)

    def dw(filter_size, knob):
        _hip.set_knob("DW_FULL", knob)
        try:
            conv = scn.SubmanifoldConvolution(3, 128, 128, filter_size, False).to(DEV)
            xx = scn.SparseConvNetTensor()
            xx.metadata, xx.spatial_size, xx.features = x.metadata, x.spatial_size, f_t
            y = conv(xx)
            with torch.autograd.set_multithreading_enabled(False):
                y.features.backward(g_t)
            return conv.weight.grad.detach().clone(), _hip.load().aabr_conv_last_variant().decode()
        finally:
            _hip.set_knob("DW_FULL", None)

    a, va = dw(3, None)
    b, vb = dw(3, 0)
    assert "k_conv_dw_full" in va and "k_conv_dw_pairs" in vb, (va, vb)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-5 * scale
    c, vc = dw(1, None)
    assert "k_conv_dw_pairs" in vc, vc

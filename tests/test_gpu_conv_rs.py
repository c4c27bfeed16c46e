"""The row-stationary bf16 convolution kernel (csrc/conv_rs.hip: k_build_rs + k_conv_rsq) through the C ABI against
the oracle fed with the SAME bf16-rounded features and weights (products are exact in fp32 either way; what differs
is the fp32 accumulation order and the final rounding to bf16).  Covers the compiled stream itself (permutation,
partner table, group sets -- integers, exact), forward and input-gradient forms, submanifold / strided / transposed
rule books, 64 and 128 input planes, one and two column slabs, several units per workgroup, ragged last units, a
unit without any rule, bias, and bit-reproducibility."""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _dev_build_only():
    """the row-stationary kernels were measured slower than the LDS-tile kernel (profiles/r03_conv_rs_ab.txt) and are
    compiled into `make DEV=1` builds only; the release library answers their entry points with an error"""
    import _hip
    if not (_hip.load().aabr_build_flags() & 1):
        lib = _hip.load()
        assert lib.aabr_conv_rs_unit_rows(128, 128, 100000, 100000, 27) == 0
        assert lib.aabr_build_rs(None, 0, 27, 64, None, None) != 0 and b"DEV=1" in lib.aabr_last_error()
        pytest.skip("release build: A/B kernels not compiled in (make DEV=1)")
DEV = "cuda:0"


def _scn():
    import sparseconvnet as scn
    return scn


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _scene(rng, n, size, batch):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64)


def _variant():
    import _hip
    return _hip.load().aabr_conv_last_variant().decode()


def _check_stream(words, table, V, vol, U):
    """the compiled stream against a numpy restatement of k_build_rs's contract"""
    nun = (V + U - 1) // U
    w = words.cpu().numpy()
    hdr = w[:nun * 32].reshape(nun, 32)
    perm = w[nun * 32:nun * (32 + U)].reshape(nun, U)
    desc = w[nun * (32 + U):nun * (32 + U + vol * 256)].reshape(nun, vol * 4, 4, 16)
    tab = table.cpu().numpy()
    for u in range(nun):
        rows = np.arange(u * U, min(V, (u + 1) * U))
        p = perm[u]
        live = p[p >= 0]
        assert sorted(live.tolist()) == rows.tolist()                       # a permutation of the unit's rows
        assert (p[len(rows):] == -1).all()
        masks = (tab[:, live] >= 0).astype(np.int64)
        mval = (masks << np.arange(vol)[:, None]).sum(0)
        assert (np.diff(mval) >= 0).all()                                   # sorted by offset mask ...
        same = np.diff(mval) == 0
        assert (np.diff(live)[same] > 0).all()                              # ... stable inside a mask class
        n = hdr[u, 0]
        ks = hdr[u, 1:n + 1] >> 16
        bits = hdr[u, 1:n + 1] & 0xffff
        assert (np.diff(ks) > 0).all()
        want = {}
        for k in range(vol):          # quads (4 groups = 64 slots) with at least one partner at offset k
            b = 0
            for q in range((len(live) + 63) // 64):
                if (tab[k, live[q * 64:(q + 1) * 64]] >= 0).any():
                    b |= 1 << q
            if b:
                want[k] = b
        assert dict(zip(ks.tolist(), bits.tolist())) == want
        # step descriptors: per active offset one step per active quad, its four groups as four items of 16 rows
        step = 0
        for k, b in zip(ks.tolist(), bits.tolist()):
            for q in range(4):
                if not (b >> q & 1):
                    continue
                for i in range(4):
                    want_e = np.full(16, -1, np.int64)
                    pr = p[q * 64 + i * 16:q * 64 + (i + 1) * 16]
                    if len(pr):
                        want_e[:len(pr)][pr >= 0] = tab[k, pr[pr >= 0]]
                    np.testing.assert_array_equal(desc[u, step, i], want_e)
                step += 1
        assert hdr[u, 31] == step


@pytest.mark.parametrize("nIn,nOut,npts,U,bias", [(128, 128, 2500, 64, False), (64, 64, 3000, 96, True),
                                                  (64, 128, 700, 32, False), (128, 256, 1800, 48, True),
                                                  (128, 64, 130, 16, False), (128, 128, 5000, 192, False)])
def test_rs_submanifold_forward_and_input_gradient(nIn, nOut, npts, U, bias):
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(nIn + 3 * nOut + npts)
    coords = _scene(rng, npts, (14, 12, 6), 2)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((npts, 1), np.float32))])
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    ga, V, vol = tb.out, tb.V_out, tb.vol
    il = O.input_layer(coords, np.zeros((npts, 1), np.float32), 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    words = ga.rs_stream(U)
    assert words.numel() == lib.aabr_rs_words(V, vol, U)
    _check_stream(words, ga.table, V, vol, U)
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    b = (rng.standard_normal(nOut)).astype(np.float32) if bias else None
    bd = _t(b) if bias else None
    f = torch.as_tensor(rng.standard_normal((V, nIn)).astype(np.float32)).to(DEV).bfloat16()
    out = torch.full((V, nOut), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rs_bf16(ptr(f), nIn, V, ptr(out), nOut, V, ptr(words), U, vol, ptr(bd), 0, ptr(pf),
                                        stream()))
    assert _variant().startswith("k_conv_rsq<"), _variant()
    ref, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, V, b)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()                                            # every output row written
    np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).max())
    # input-gradient form: transposed pack, mirrored offsets, nOut -> nIn
    if nOut in (64, 128) and nIn % 64 == 0:
        g = torch.as_tensor(rng.standard_normal((V, nOut)).astype(np.float32)).to(DEV).bfloat16()
        d_in = torch.empty((V, nIn), dtype=torch.bfloat16, device=DEV)
        check(lib.aabr_conv_forward_rs_bf16(ptr(g), nOut, V, ptr(d_in), nIn, V, ptr(words), U, vol, None, 1 | 2,
                                            ptr(pt), stream()))
        dref, _, _ = O.conv_bwd(np.zeros((V, nIn), np.float32), g.float().cpu().numpy(), Wr, rb, want_bias=False)
        np.testing.assert_allclose(d_in.float().cpu().numpy(), dref, rtol=2 ** -7, atol=2 ** -7 * np.abs(dref).max())
    out2 = torch.empty_like(out)
    check(lib.aabr_conv_forward_rs_bf16(ptr(f), nIn, V, ptr(out2), nOut, V, ptr(words), U, vol, ptr(bd), 0, ptr(pf),
                                        stream()))
    assert torch.equal(out, out2)                                            # same launch, same bits


@pytest.mark.parametrize("fs,st,nIn,nOut,U", [([2, 2, 2], [2, 2, 2], 64, 128, 64), ([1, 1, 8], [1, 1, 1], 128, 128, 32),
                                              ([2, 2, 2], [2, 2, 2], 128, 128, 160)])
def test_rs_strided_and_transposed_books(fs, st, nIn, nOut, U):
    """Convolution (tb.out: several partners per output row, some output rows of a unit without any at an offset)
    and Deconvolution over the same book (tb.inn: exactly one partner per fine row -> single-offset groups)"""
    import _hip
    from _hip import ptr, stream, check
    scn = _scn()
    lib = _hip.load()
    rng = np.random.default_rng(sum(fs) + nIn + nOut)
    coords = _scene(rng, 4000, (16, 16, 8), 2)
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(np.zeros((4000, 1), np.float32))])
    isz = x.spatial_size
    osz = (isz - torch.LongTensor(fs)) // torch.LongTensor(st) + 1
    tb = x.metadata.getRuleBook(isz, osz, torch.LongTensor(fs), torch.LongTensor(st))
    il = O.input_layer(coords, np.zeros((4000, 1), np.float32), 4)
    rb, oc = O.convolution_rules(il["coords"], fs, st, osz.tolist())
    vol = tb.vol
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    # Convolution forward: fine -> coarse over tb.out
    f = torch.as_tensor(rng.standard_normal((tb.V_in, nIn)).astype(np.float32)).to(DEV).bfloat16()
    out = torch.full((tb.V_out, nOut), float("nan"), dtype=torch.bfloat16, device=DEV)
    w_out = tb.out.rs_stream(U)
    _check_stream(w_out, tb.out.table, tb.V_out, vol, U)
    check(lib.aabr_conv_forward_rs_bf16(ptr(f), nIn, tb.V_in, ptr(out), nOut, tb.V_out, ptr(w_out), U, vol, None, 0,
                                        ptr(pf), stream()))
    ref, _ = O.conv_fwd(f.float().cpu().numpy(), Wr, rb, tb.V_out, None)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).max())
    # its input gradient = the Deconvolution-shaped gather over tb.inn with the transposed pack (coarse -> fine)
    if nOut in (64, 128):
        g = torch.as_tensor(rng.standard_normal((tb.V_out, nOut)).astype(np.float32)).to(DEV).bfloat16()
        d_in = torch.full((tb.V_in, nIn), float("nan"), dtype=torch.bfloat16, device=DEV)
        w_in = tb.inn.rs_stream(U)
        _check_stream(w_in, tb.inn.table, tb.V_in, vol, U)
        check(lib.aabr_conv_forward_rs_bf16(ptr(g), nOut, tb.V_out, ptr(d_in), nIn, tb.V_in, ptr(w_in), U, vol, None,
                                            1, ptr(pt), stream()))
        dref, _, _ = O.conv_bwd(np.zeros((tb.V_in, nIn), np.float32), g.float().cpu().numpy(), Wr, rb,
                                want_bias=False)
        got = d_in.float().cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, dref, rtol=2 ** -7, atol=2 ** -7 * np.abs(dref).max())


def test_rs_units_without_rules_and_dispatch():
    """a rule table whose middle unit has no rule at all (possible for strided books) still gets its rows written
    (zeros + bias); the dispatch takes the kernel only for bf16-sized launches that fill the chip"""
    import _hip
    from _hip import ptr, stream, check
    lib = _hip.load()
    V, vol, U, nIn, nOut = 100, 8, 32, 64, 64
    rng = np.random.default_rng(0)
    tab = rng.integers(-1, V, (vol, V)).astype(np.int32)
    tab[rng.random((vol, V)) < 0.6] = -1
    tab[:, 32:64] = -1
    table = _t(tab)
    words = torch.empty(lib.aabr_rs_words(V, vol, U), dtype=torch.int32, device=DEV)
    check(lib.aabr_build_rs(ptr(table), V, vol, U, ptr(words), stream()))
    _check_stream(words, table, V, vol, U)
    assert words[32].item() == 0 and words[32 + 31].item() == 0              # unit 1: no active offset, no step
    W = (rng.standard_normal((vol, 1, nIn, nOut)) * 0.1).astype(np.float32)
    Wd = _t(W)
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, nIn, nOut))
    pf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    pt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(Wd), vol, nIn, nOut, ptr(pf), ptr(pt), stream()))
    f = torch.as_tensor(rng.standard_normal((V, nIn)).astype(np.float32)).to(DEV).bfloat16()
    b = _t(rng.standard_normal(nOut).astype(np.float32))
    out = torch.full((V, nOut), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.aabr_conv_forward_rs_bf16(ptr(f), nIn, V, ptr(out), nOut, V, ptr(words), U, vol, ptr(b), 0, ptr(pf),
                                        stream()))
    Wr = Wd.bfloat16().float().cpu().numpy().reshape(vol, nIn, nOut)
    ff = f.float().cpu().numpy()
    ref = np.tile(b.cpu().numpy(), (V, 1)).astype(np.float64)
    for k in range(vol):
        rows = np.nonzero(tab[k] >= 0)[0]
        ref[rows] += ff[tab[k, rows]].astype(np.float64) @ Wr[k]
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=2 ** -7 * np.abs(ref).max())
    np.testing.assert_array_equal(out[32:64].float().cpu().numpy(),
                                  np.tile(b.bfloat16().float().cpu().numpy(), (32, 1)))
    # dispatch: off unless the CONV_RS knob says 1 (the LDS-tile kernel is faster: profiles/r03_conv_rs_ab.txt);
    # unsupported shapes say 0 either way
    assert lib.aabr_conv_rs_unit_rows(128, 128, 84077, 84077, 27) == 0
    _hip.set_knob("CONV_RS", 1)
    try:
        assert lib.aabr_conv_rs_unit_rows(128, 128, 84077, 84077, 27) in range(16, 257, 16)
        assert lib.aabr_conv_rs_unit_rows(32, 64, 300000, 300000, 27) == 0
        assert lib.aabr_conv_rs_unit_rows(128, 96, 300000, 300000, 27) == 0
    finally:
        _hip.set_knob("CONV_RS", None)

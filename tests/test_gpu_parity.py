"""Parity of the HIP path (through the reference-shaped Python boundary and the C ABI) against
the CPU oracle, on a real MI355X.  Integer structures are compared bit-exactly; floating point
within the tolerance written at each assert."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import ref_net
import synth_scenes as S

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _scn():
    import sparseconvnet as scn
    return scn


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _rand_scene(rng, n, size, batch, C):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64), rng.standard_normal((n, C)).astype(np.float32)


def _input(scn, coords, feats, spatial, mode=4):
    layer = scn.InputLayer(3, list(spatial), mode=mode)
    f = _t(feats).requires_grad_(True)
    return layer([_t(coords), f]), f


# ------------------------------------------------------------------------------------ input layer
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_input_layer_sites_rules_and_features_exact(mode):
    scn = _scn()
    rng = np.random.default_rng(10 + mode)
    coords, feats = _rand_scene(rng, 5000, (12, 10, 6), 3, 7)  # heavy duplication (~3 pts / voxel)
    x, f = _input(scn, coords, feats, (16, 16, 8), mode)
    ref = O.input_layer(coords, feats, mode)
    md = x.metadata
    assert md.input["V"] == ref["V"]
    np.testing.assert_array_equal(md.getSpatialLocations(x.spatial_size).numpy(), ref["coords"])
    np.testing.assert_array_equal(md.input["point_site"].cpu().numpy(), ref["point_voxel"])
    hdr, rules = md.inputLayerRuleBook()
    assert hdr == [mode, ref["max_active"], 5000, ref["V"]]
    np.testing.assert_array_equal(rules.cpu().numpy(), ref["rules"])
    # same operation order, no FMA contraction -> bit-exact fp32
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"])
    g = rng.standard_normal(ref["out"].shape).astype(np.float32)
    x.features.backward(_t(g))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), O.input_layer_bwd(ref, g))


def test_input_layer_edge_cases():
    scn = _scn()
    import _hip
    # 3-column coordinates (no batch column), single point, all points in one voxel
    for coords in (np.array([[3, 4, 5]], np.int64), np.tile(np.array([[7, 7, 7]], np.int64), (1000, 1)),
                   np.array([[0, 0, 0], [65534, 65534, 65534], [0, 0, 0]], np.int64)):
        feats = np.arange(coords.shape[0] * 2, dtype=np.float32).reshape(-1, 2)
        x, _ = _input(scn, coords, feats, (70000, 70000, 70000), 4)
        ref = O.input_layer(coords, feats, 4)
        np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"])
        np.testing.assert_array_equal(x.get_spatial_locations().numpy()[:, :3], ref["coords"][:, :3])
    # empty input
    x, _ = _input(scn, np.zeros((0, 4), np.int64), np.zeros((0, 3), np.float32), (8, 8, 8), 4)
    assert tuple(x.features.shape) == (0, 3) and x.get_spatial_locations().shape == (0, 4)
    # out-of-range coordinates are rejected loudly
    with pytest.raises(_hip.AabrError):
        _input(scn, np.array([[1, 2, -3, 0]], np.int64), np.zeros((1, 2), np.float32), (8, 8, 8), 4)
    with pytest.raises(_hip.AabrError):
        _input(scn, np.array([[1, 2, 70000, 0]], np.int64), np.zeros((1, 2), np.float32), (8, 8, 8), 4)


# ------------------------------------------------------------------------------------ rule books
def _pairs_set(p):
    return {(int(a), int(b)) for a, b in p}


def test_submanifold_rulebook_exact():
    scn = _scn()
    rng = np.random.default_rng(20)
    coords, feats = _rand_scene(rng, 4000, (20, 18, 9), 2, 3)
    x, _ = _input(scn, coords, feats, (32, 32, 16), 3)
    ref_il = O.input_layer(coords, feats, 3)
    for fs in ([3, 3, 3], [1, 1, 1], [3, 1, 5]):
        tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor(fs))
        rb = O.submanifold_rules(ref_il["coords"], fs)
        np.testing.assert_array_equal(np.array(tb.rule_counts()), rb.counts)
        got = scn.SCN.Metadata_3.tableToRuleBook(tb.out.table)
        for k in range(rb.vol):  # same order too: ascending output row within an offset
            np.testing.assert_array_equal(got[k].cpu().numpy(), rb.pairs(k))


@pytest.mark.parametrize("fs,st", [([2, 2, 2], [2, 2, 2]), ([3, 3, 3], [2, 2, 2]), ([1, 1, 8], [1, 1, 1]),
                                   ([4, 4, 4], [2, 2, 2])])
def test_strided_rulebook_and_output_sites_exact(fs, st):
    scn = _scn()
    rng = np.random.default_rng(21)
    size = np.array([17, 15, 8]) if fs != [1, 1, 8] else np.array([12, 12, 8])
    if fs == [2, 2, 2]:
        size = np.array([16, 16, 8])
    if fs == [4, 4, 4]:
        size = np.array([18, 18, 10])
    coords, feats = _rand_scene(rng, 3000, tuple(size), 3, 3)
    x, _ = _input(scn, coords, feats, tuple(size), 3)
    ref_il = O.input_layer(coords, feats, 3)
    osz = (size - np.array(fs)) // np.array(st) + 1
    tb = x.metadata.getRuleBook(x.spatial_size, torch.LongTensor(osz), torch.LongTensor(fs), torch.LongTensor(st))
    rb, oc = O.convolution_rules(ref_il["coords"], fs, st, osz)
    # output sites: same set, same (insertion) order, batch-contiguous
    np.testing.assert_array_equal(x.metadata.getSpatialLocations(torch.LongTensor(osz)).numpy(), oc)
    np.testing.assert_array_equal(np.array(tb.rule_counts()), rb.counts)
    t_out = scn.SCN.Metadata_3.tableToRuleBook(tb.out.table)   # (in, out) pairs
    t_in = scn.SCN.Metadata_3.tableToRuleBook(tb.inn.table)    # (out, in) pairs
    for k in range(rb.vol):
        want = _pairs_set(rb.pairs(k))
        assert _pairs_set(t_out[k].cpu().numpy()) == want
        assert {(b, a) for a, b in _pairs_set(t_in[k].cpu().numpy())} == want


# ------------------------------------------------------------------------------------ convolutions
CONV_SHAPES = [(9, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (16, 8), (5, 40), (128, 256)]


@pytest.mark.parametrize("nIn,nOut", CONV_SHAPES)
def test_submanifold_conv_forward_backward(nIn, nOut):
    scn = _scn()
    rng = np.random.default_rng(nIn * 1000 + nOut)
    coords, feats = _rand_scene(rng, 2500, (14, 12, 6), 2, nIn)
    x, f = _input(scn, coords, feats, (16, 16, 8), 4)
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, nIn == 16).to(DEV)
    if nIn == 16:
        conv.bias.data.normal_()
    y = conv(x)
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = conv.weight.detach().cpu().numpy().reshape(27, nIn, nOut)
    bias = conv.bias.detach().cpu().numpy() if nIn == 16 else None
    ref, macs = O.conv_fwd(il["out"], W, rb, il["V"], bias)
    scale = np.abs(ref).max()
    # fp32 MFMA (k-ordered fmaf chain) vs double-accumulated oracle
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * scale * nIn)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, db = O.conv_bwd(il["out"], g, W, rb, want_bias=bias is not None)
    d_feats = O.input_layer_bwd(il, d_in)
    np.testing.assert_allclose(f.grad.cpu().numpy(), d_feats, rtol=1e-4, atol=2e-6 * np.abs(d_feats).max() * nOut)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(27, nIn, nOut), dW, rtol=1e-4,
                               atol=1e-5 * np.abs(dW).max())
    if bias is not None:
        np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4)


def test_multiply_add_counter_matches_reference_definition():
    scn = _scn()
    rng = np.random.default_rng(31)
    coords, feats = _rand_scene(rng, 1500, (10, 10, 6), 1, 8)
    scn.forward_pass_multiplyAdd_count = 0
    x, _ = _input(scn, coords, feats, (16, 16, 8), 4)
    conv = scn.SubmanifoldConvolution(3, 8, 16, 3, False).to(DEV)
    conv(x)
    il = O.input_layer(coords, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    assert scn.forward_pass_multiplyAdd_count == rb.total * 8 * 16  # CPU/Convolution.cpp:141


@pytest.mark.parametrize("fs,st,nIn,nOut", [([2, 2, 2], [2, 2, 2], 32, 64), ([2, 2, 2], [2, 2, 2], 64, 64),
                                            ([1, 1, 8], [1, 1, 1], 128, 128), ([3, 3, 3], [2, 2, 2], 16, 24)])
def test_strided_conv_and_deconv_forward_backward(fs, st, nIn, nOut):
    scn = _scn()
    rng = np.random.default_rng(40 + nIn)
    size = np.array([16, 16, 8]) if fs != [3, 3, 3] else np.array([17, 17, 9])
    coords, feats = _rand_scene(rng, 2500, tuple(size), 2, nIn)
    x, f = _input(scn, coords, feats, tuple(size), 4)
    conv = scn.Convolution(3, nIn, nOut, fs, st, False).to(DEV)
    dec = scn.Deconvolution(3, nOut, nIn, fs, st, False).to(DEV)
    y = conv(x)
    z = dec(y)
    assert z.spatial_size.tolist() == list(size)
    il = O.input_layer(coords, feats, 4)
    osz = (size - np.array(fs)) // np.array(st) + 1
    rb, oc = O.convolution_rules(il["coords"], fs, st, osz)
    Wc = conv.weight.detach().cpu().numpy().reshape(rb.vol, nIn, nOut)
    Wd = dec.weight.detach().cpu().numpy().reshape(rb.vol, nOut, nIn)
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


# ------------------------------------------------------------------------------------ batch norm
def _bn_exact(x, w, b, eps, leak, g):
    """BatchNorm(+leaky ReLU) forward and backward in fp64 from the same fp32 inputs: the yardstick both the device
    (fp64 partial sums, fp32 affine maths) and the oracle (the reference's sequential fp32 sums,
    SCN/CPU/BatchNormalization.cpp:20-48,85-106) are measured against"""
    x = x.astype(np.float64)
    n = x.shape[0]
    mean = x.mean(0)
    var = ((x - mean) ** 2).sum(0) / n
    invstd = (var + eps) ** -0.5
    y = (x - mean) * invstd * w + b
    out = np.where(y > 0, y, y * leak)
    d = np.where(y > 0, g, g * leak).astype(np.float64)
    db = d.sum(0)
    dot = ((x - mean) * d).sum(0)
    dw = dot * invstd
    d_in = (d - db / n - (x - mean) * (dot * invstd * invstd / n)) * (invstd * w)
    return out, mean, var, d_in, dw, db, y


@pytest.mark.parametrize("planes,leak,npts", [(32, 0.0, 6000), (64, 0.333, 1800), (128, 0.0, 1500), (9, 0.1, 6000),
                                              (256, 0.0, 600), (64, 0.2, 60000), (128, 0.0, 30000), (9, 0.0, 30000),
                                              (128, 0.1, 6000)])
def test_batchnorm_forward_backward(planes, leak, npts):
    """<= 1800 points (<= 2048 sites): the one-launch small-map kernels; 6000 / 30000 / 60000 points: statistics pass +
    finalize + apply.  Which side is closer to exact: the device keeps fp64 partial sums, the reference (and the
    oracle, bit-pinned to it) sums sequentially in fp32 -- measured against the fp64 yardstick the device's error is
    the smaller one (asserted below), and the device-vs-oracle tolerances are set from the ORACLE's measured
    distance to exact (~1e-5 relative on the statistics at these sizes), not from a guess."""
    scn = _scn()
    rng = np.random.default_rng(50 + planes + npts)
    big = npts > 8192
    coords, feats = _rand_scene(rng, npts, (64, 64, 16) if big else (24, 24, 8), 2, planes)
    feats = (feats * 1.7 + 0.4).astype(np.float32)
    x, f = _input(scn, coords, feats, (64, 64, 16) if big else (32, 32, 8), 4)
    bn = scn.BatchNormLeakyReLU(planes, momentum=0.95, leakiness=leak).to(DEV)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    w, b = bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()
    y = bn(x)
    il = O.input_layer(coords, feats, 4)
    V = il["V"]
    assert (V > 8192) == big
    out, sm, si, rm, rv = O.bn_fwd(il["out"], w, b, np.zeros(planes), np.ones(planes), 1e-4, 0.95, True, leak)
    g = rng.standard_normal(out.shape).astype(np.float32)
    y.features.backward(_t(g))
    yd = y.features.detach().cpu().numpy()
    ex_out, ex_mean, ex_var, ex_din, ex_dw, ex_db, ex_y = _bn_exact(il["out"], w, b, 1e-4, leak, g)
    scale = np.abs(ex_out).max()
    # forward: device within fp32 rounding of the exact result, and closer to it than the sequential-fp32 oracle
    err_dev, err_orc = np.abs(yd - ex_out).max() / scale, np.abs(out - ex_out).max() / scale
    assert err_dev <= 4e-7, err_dev
    assert err_dev <= err_orc + 1e-7, (err_dev, err_orc)
    np.testing.assert_allclose(yd, out, rtol=0, atol=(err_orc + 4e-7) * scale)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), 0.05 * ex_mean, rtol=2e-7, atol=1e-8)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), 0.95 + 0.05 * ex_var * V / (V - 1), rtol=3e-7)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), rm, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), rv, rtol=2e-5)
    # activation masks come from the forward output's sign; evaluate the backward yardsticks on the
    # device's own forward output so an activation within rounding distance of 0 cannot flip it
    assert ((yd > 0) != (out > 0)).sum() <= 1e-5 * out.size + 2
    flip = (yd > 0) != (ex_y > 0)
    assert flip.sum() <= 1e-5 * out.size + 2
    gscale = np.abs(ex_din).max()
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ex_db, rtol=1e-6, atol=1e-6 * np.abs(g).sum(0).max())
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ex_dw, rtol=1e-5, atol=2e-6 * np.abs(ex_dw).max() + 1e-5)
    d_feats_exact = O.input_layer_bwd(il, ex_din.astype(np.float32))
    got = f.grad.cpu().numpy()
    d_in, dw, db, _ = O.bn_bwd(il["out"], yd, g, sm, si, w, leak)
    d_feats = O.input_layer_bwd(il, d_in)
    err_dev_b = np.abs(got - d_feats_exact).max() / gscale
    err_orc_b = np.abs(d_feats - d_feats_exact).max() / gscale
    if flip.sum() == 0:
        assert err_dev_b <= 2e-6, err_dev_b
        assert err_dev_b <= err_orc_b + 5e-7, (err_dev_b, err_orc_b)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), dw, rtol=1e-4, atol=1e-4 * np.abs(dw).max())
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4 * np.abs(db).max())
    np.testing.assert_allclose(got, d_feats, rtol=0, atol=(err_orc_b + 2e-6) * gscale)
    # eval mode uses the running statistics (BatchNormalization.cpp:42-47)
    bn.eval()
    ye = bn(x).features.detach().cpu().numpy()
    oe, *_ = O.bn_fwd(il["out"], w, b, bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), 1e-4, 0.95,
                      False, leak)
    np.testing.assert_allclose(ye, oe, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------ 2-stage slice
def _two_stage_modules(scn, c_in):
    torch.manual_seed(0)
    m = dict(conv1=scn.SubmanifoldConvolution(3, c_in, 32, 3, False), bn1=scn.BatchNormLeakyReLU(32, momentum=0.95, leakiness=0),
             conv2=scn.SubmanifoldConvolution(3, 32, 32, 3, False), bn2=scn.BatchNormLeakyReLU(32, momentum=0.95, leakiness=0),
             conv3=scn.SubmanifoldConvolution(3, 32, 32, 3, False))
    for v in m.values():
        v.to(DEV)
    return m


def _run_two_stage(scn, m, locs, feats, g=None):
    layer = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)
    f = _t(feats).requires_grad_(True)
    x0 = layer([_t(locs), f])
    x1 = m["conv1"](x0)
    y1 = m["bn1"](x1)
    x2 = m["conv2"](y1)
    y2 = m["bn2"](x2)
    x3 = m["conv3"](y2)
    out = scn.add_feature_planes([x1, x3])
    if g is not None:
        out.features.backward(_t(g))
    acts = dict(x0=x0, x1=x1, y1=y1, x2=x2, y2=y2, x3=x3)
    return out, f, {k: v.features.detach().cpu().numpy() for k, v in acts.items()}


def _relerr(got, ref):
    return float(np.linalg.norm(got.astype(np.float64) - ref) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30))


@pytest.mark.parametrize("npts,batch", [(3000, 2), (80000, 1)])
def test_two_stage_slice_matches_oracle(npts, batch):
    """BASELINE.json configs[1] (and a small batched variant): scene -> voxel scatter -> 2-stage
    submanifold backbone, forward and backward, against the oracle composition."""
    scn = _scn()
    locs, feats = S.make_batch(batch, npts, 0, 20)
    m = _two_stage_modules(scn, 9)
    P = lambda t: t.detach().cpu().numpy()
    W1, W2, W3 = (P(m[k].weight).reshape(27, -1, 32) for k in ("conv1", "conv2", "conv3"))
    bn = [dict(weight=P(m[k].weight), bias=P(m[k].bias), running_mean=np.zeros(32), running_var=np.ones(32))
          for k in ("bn1", "bn2")]
    c = ref_net.two_stage_forward(locs, feats, W1, W2, W3, bn[0], bn[1])
    rng = np.random.default_rng(3)
    g = rng.standard_normal(c["out"].shape).astype(np.float32)
    out, f, acts = _run_two_stage(scn, m, locs, feats, g)
    if npts == 80000:
        assert c["il"]["V"] == 66094 and c["rb"].total == 243374
    assert out.features.shape[0] == c["il"]["V"]
    np.testing.assert_array_equal(out.get_spatial_locations().numpy(), c["il"]["coords"])
    np.testing.assert_array_equal(acts["x0"], c["x0"])          # voxel means: bit-exact
    # forward chain: fp32 MFMA + fp64-partial BN statistics vs the oracle's double-accumulated
    # GEMM + sequential-fp32 BN statistics (the reference's own arithmetic)
    for k in ("x1", "y1", "x2", "y2", "x3"):
        assert _relerr(acts[k], c[k]) < 2e-4, k
    np.testing.assert_allclose(P(out.features), c["out"], rtol=2e-3, atol=2e-4 * np.abs(c["out"]).max())
    # backward: ReLU masks are taken from the forward activations, and an activation within
    # rounding distance of 0 may fall on the other side; so the oracle backward is evaluated on
    # the device's own forward activations (identical masks), which isolates the backward kernels
    flips = int(((acts["y1"] > 0) != (c["y1"] > 0)).sum() + ((acts["y2"] > 0) != (c["y2"] > 0)).sum())
    assert flips <= 1e-5 * acts["y1"].size + 2
    cg = dict(c)
    cg.update(acts)
    r = ref_net.two_stage_backward(cg, g, W1, W2, W3, bn[0], bn[1])
    for name, ref in (("conv1", r["dW1"]), ("conv2", r["dW2"]), ("conv3", r["dW3"])):
        got = P(m[name].weight.grad).reshape(ref.shape)
        assert _relerr(got, ref) < 2e-4, name
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
    for name, ref in (("bn1.weight", r["dbn1w"]), ("bn1.bias", r["dbn1b"]), ("bn2.weight", r["dbn2w"]),
                      ("bn2.bias", r["dbn2b"])):
        mod, attr = name.split(".")
        got = P(getattr(m[mod], attr).grad)
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    assert _relerr(P(f.grad), r["d_feats"]) < 5e-4
    np.testing.assert_allclose(P(f.grad), r["d_feats"], rtol=5e-3, atol=5e-4 * np.abs(r["d_feats"]).max())


def test_full_size_properties():
    """size-independent properties at the BASELINE configs[4]-style scale (1.5 M points @ 2 cm):
    bit-reproducibility, linearity of the convolution, restoration of the fine grid."""
    scn = _scn()
    locs, feats = S.make_scene(1500000, 0, 50)
    layer = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)
    conv = scn.SubmanifoldConvolution(3, 9, 32, 3, False).to(DEV)
    down = scn.Convolution(3, 32, 32, 2, 2, False).to(DEV)
    up = scn.Deconvolution(3, 32, 32, 2, 2, False).to(DEV)
    with torch.no_grad():
        l, f = _t(locs), _t(feats)
        x = layer([l, f])
        V = x.features.shape[0]
        il = O.input_layer(locs, None, 4)
        assert V == il["V"]
        np.testing.assert_array_equal(x.get_spatial_locations().numpy()[:, :3], il["coords"][:, :3])
        y1 = conv(x).features
        x2 = layer([l, f])
        y2 = conv(x2).features
        assert torch.equal(y1, y2)  # deterministic: no atomics in the accumulation
        # linearity: conv(a*x) == a*conv(x) up to rounding
        xs = scn.SparseConvNetTensor(x.features * 2.0, x.metadata, x.spatial_size)
        torch.testing.assert_close(conv(xs).features, y1 * 2.0, rtol=1e-6, atol=1e-6)
        tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
        rb = O.submanifold_rules(il["coords"], [3, 3, 3])
        np.testing.assert_array_equal(np.array(tb.rule_counts()), rb.counts)
        # checksum of the rule table: sum over rules of (in + 3*out) per offset
        t = tb.out.table.to(torch.int64)
        rows = torch.arange(V, device=DEV, dtype=torch.int64)[None, :]
        chk = torch.where(t >= 0, t + 3 * rows, torch.zeros_like(t)).sum(1).cpu().numpy()
        want = np.array([(rb.pairs(k)[:, 0].astype(np.int64) + 3 * rb.pairs(k)[:, 1].astype(np.int64)).sum()
                         for k in range(27)])
        np.testing.assert_array_equal(chk, want)
        xc = scn.SparseConvNetTensor(y1, x.metadata, x.spatial_size)
        z = up(down(xc))
        assert z.features.shape[0] == V and z.spatial_size.tolist() == list(S.FULL_SCALE)


def test_fpn_net_forward_backward_runs_and_is_consistent():
    """default 9-scale backbone on a small scene: map sizes, site consistency with the oracle's
    strided geometry, finite gradients for every parameter."""
    from test_cabi_and_host import default_fpn
    scn = _scn()
    torch.manual_seed(1)
    net = default_fpn().to(DEV)
    locs, feats = S.make_batch(2, 20000, 5, 20)
    rpn_maps, roi_maps = net([_t(locs), _t(feats)])
    assert len(rpn_maps) == 6 and len(roi_maps) == 4
    il = O.input_layer(locs, None, 4)
    sites = il["coords"]
    size = np.array(S.FULL_SCALE)
    expect = {}
    for _ in range(8):
        osz = (size - 2) // 2 + 1
        rb, sites = O.convolution_rules(sites, [2, 2, 2], [2, 2, 2], osz)
        size = osz
        expect[tuple(size)] = sites
    for mp in rpn_maps[:3] + roi_maps:
        key = tuple(mp.spatial_size.tolist())
        assert mp.features.shape == (expect[key].shape[0], 128)
        np.testing.assert_array_equal(mp.get_spatial_locations().numpy(), expect[key])
    loss = sum(m.features.square().mean() for m in rpn_maps)
    loss.backward()
    for n, p in net.named_parameters():
        # not reached from the selected maps: fpn_scales_from_top = [4,3,2,1] uses ups[1..4] only, so the
        # four finest up-path stages (and their lateral shortcuts) and the dropped 2-D head get no gradient
        dead = ("linear", "layers_out", "convs_pro2d.3") + tuple(
            "%s.%d." % (a, i) for a, r in (("m_shortcuts", range(0, 4)), ("m_ups", range(4, 8)),
                                           ("m_mergeds", range(4, 8))) for i in r)
        if n.startswith(dead) or (n + ".").startswith(dead):
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert torch.isfinite(loss)


def test_fpn_net_multiply_add_count_matches_reference_counter():
    """BASELINE.md §2 "Derived": the reference's own counter (forward_pass_multiplyAdd_count) reports
    ~21.3 G MACs for one default-backbone forward on S80k @ 5 cm.  Same counter definition here
    (sum over conv layers of rules x nIn x nOut), evaluated lazily from the device rule counts."""
    from test_cabi_and_host import default_fpn
    scn = _scn()
    torch.manual_seed(0)
    net = default_fpn().to(DEV)
    locs, feats = S.make_batch(1, 80000, 0, 20)
    scn.forward_pass_multiplyAdd_count = 0
    with torch.no_grad():
        rpn_maps, _ = net([_t(locs), _t(feats)])
    macs = float(scn.forward_pass_multiplyAdd_count)
    assert abs(macs - 21.3e9) / 21.3e9 < 0.01, macs
    assert rpn_maps[0].metadata.input["V"] == 66094


# ------------------------------------------------------------------------------------ IoU / NMS
def test_rotate_iou_matrix_vs_oracle_and_golden(golden_dir):
    from second.core.non_max_suppression.nms_gpu import rotate_iou_gpu_eval
    g = np.load(os.path.join(golden_dir, "iou_golden.npz"))
    # device cosf/sinf/sqrtf may differ from the host's by an ulp: 1e-5 absolute (SURVEY §8c quotes 1e-6
    # for the reference's own host-vs-device discrepancy on well-conditioned pairs)
    np.testing.assert_allclose(rotate_iou_gpu_eval(g["t_boxes"], g["t_boxes"]), g["t_iou"], atol=1e-5)
    np.testing.assert_allclose(rotate_iou_gpu_eval(g["hw_boxes"], g["hw_boxes"], -1), g["hw_iou_c-1"], atol=1e-5)
    np.testing.assert_allclose(rotate_iou_gpu_eval(g["lab_gt"], g["lab_anchor"], 6), g["lab_iou_c6"], atol=1e-5)
    b7, _ = S.make_nms_boxes(700, 11)
    b5 = np.ascontiguousarray(b7[:, [0, 1, 3, 4, 6]])
    for crit in (-1, 0, 1, 2, 6):
        got = rotate_iou_gpu_eval(b5[:300], b5[300:], crit)
        np.testing.assert_allclose(got, O.rotate_iou_eval(b5[:300], b5[300:], crit), atol=2e-5)
    assert rotate_iou_gpu_eval(np.zeros((0, 5), np.float32), b5).shape == (0, 700)


def test_boxes_iou_3d_with_clamps():
    from utils3d.rotate_nms_3d_torch import boxes_iou_3d
    import _nms
    b7, _ = S.make_nms_boxes(400, 12)
    aug = {"target_Y": 0.3, "target_Z": 0.0, "anchor_Y": 0.0, "anchor_Z": 0.0}
    got = boxes_iou_3d(_t(b7[:40]), _t(b7[40:]), aug, 6, flag="rpn_label_generation").cpu().numpy()
    np.testing.assert_allclose(got, O.boxes_iou_3d(b7[:40], b7[40:], (0.3, 0, 0, 0), 6, True), atol=2e-5)
    full = _nms.boxes_iou_3d(_t(b7[:40]), _t(b7[40:]), (0.1, 2.5, 0.2, 2.6), -1, only_xy=False).cpu().numpy()
    np.testing.assert_allclose(full, O.boxes_iou_3d(b7[:40], b7[40:], (0.1, 2.5, 0.2, 2.6), -1, False), atol=2e-5)


@pytest.mark.parametrize("n,post,thr", [(2000, 1000, 0.5), (2000, 100, 0.3), (777, 1000, 0.7), (64, 10, 0.1),
                                        (1, 5, 0.5)])
def test_rotate_nms_3d_survivors_exact(n, post, thr):
    from second.pytorch.core.box_torch_ops import rotate_nms_3d
    b7, sc = S.make_nms_boxes(n, 13 + n)
    want = O.rotate_nms_3d(b7, sc, 2000, post, thr)
    got = rotate_nms_3d(_t(b7), _t(sc), pre_max_size=2000, post_max_size=post, iou_threshold=thr, flag="rpn_post")
    got = got.cpu().numpy()
    # Survivors are bit-exact unless some pair's IoU lies within rounding distance of the threshold
    # (device cosf/sinf vs libm).  Find such pairs from the device's own IoU matrix; if there are
    # none the keep lists must be identical, otherwise the device list must equal the oracle's
    # greedy rule applied to the device matrix and the disagreeing pairs must be razor-edge.
    import _nms
    idx = np.argsort(-sc, kind="stable")[: min(n, 2000)]
    b = b7[idx]
    iou_o = O.boxes_iou_3d(b, b)
    iou_d = _nms.boxes_iou_3d(_t(b), _t(b), (0, 0, 0, 0), -1, True).cpu().numpy()
    np.testing.assert_allclose(iou_d, iou_o, atol=2e-5)
    amb = (iou_d >= thr) != (iou_o >= thr)
    if not amb.any():
        np.testing.assert_array_equal(got, want)
    else:
        assert (np.abs(iou_o[amb] - thr) < 2e-5).all()
        keep = O.nms_from_matrix(iou_d, np.arange(len(idx), dtype=np.int32), thr)[:post]
        np.testing.assert_array_equal(got, idx[keep])


def test_nms_degenerate_inputs():
    from second.pytorch.core.box_torch_ops import rotate_nms_3d
    from maskrcnn_benchmark.structures.boxlist_ops_3d import boxlist_nms_3d
    assert rotate_nms_3d(torch.zeros(0, 7, device=DEV), torch.zeros(0, device=DEV), 2000, 100, 0.5).numel() == 0
    # identical boxes: IoU forced to 1 -> only the best-scored survives
    b = np.tile(np.array([[1, 1, 0, 0.2, 3, 2.5, 0.3]], np.float32), (130, 1))
    sc = np.linspace(0, 1, 130).astype(np.float32)
    k = rotate_nms_3d(_t(b), _t(sc), 2000, 1000, 0.5).cpu().numpy()
    assert k.tolist() == [129]
    # disjoint boxes: everything survives, in descending score order
    b2 = b.copy()
    b2[:, 0] = np.arange(130) * 10
    k = rotate_nms_3d(_t(b2), _t(sc), 2000, 1000, 0.5).cpu().numpy()
    assert k.tolist() == list(range(129, -1, -1))

    class BL(object):  # duck-typed BoxList3D
        mode = "yx_zb"

        def __init__(self, b, s):
            self.bbox3d, self.s = b, s

        def get_field(self, _):
            return self.s

        def __len__(self):
            return self.bbox3d.shape[0]

        def __getitem__(self, idx):
            return BL(self.bbox3d[idx], self.s[idx])

    b7, s = S.make_nms_boxes(500, 99)
    out = boxlist_nms_3d(BL(_t(b7), _t(s)), 0.5, [0.3, 0.3], 400, flag="rpn_post")
    bc = b7.copy()
    bc[:, 3:5] = np.maximum(bc[:, 3:5], 0.3)
    bc[:, 5] = np.maximum(bc[:, 5], 0.3)
    want = O.rotate_nms_3d(bc, s, 2000, 400, 0.5)
    np.testing.assert_array_equal(out.s.cpu().numpy(), s[want])


def test_axis_aligned_nms_matches_reference_cpu_rule():
    from maskrcnn_benchmark.layers import nms
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 100, (600, 2))
    wh = rng.uniform(5, 40, (600, 2))
    dets = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    sc = rng.random(600).astype(np.float32)
    got = nms(_t(dets), _t(sc), 0.4).cpu().numpy()
    np.testing.assert_array_equal(got, O.nms_axis_aligned(dets, sc, 0.4))


# ------------------------------------------------------------------------------------ reference-torch golden
def test_rpn_decode_kernel_vs_reference_torch_golden(golden_dir):
    """k_rpn_decode against BoxCoder3D.decode run from the reference (tests/golden/gen_box_golden.py): the 256
    golden anchors ride as base anchors of a single site at the origin, so the kernel's anchor = base."""
    import _hip
    from _hip import ptr, stream, check
    g = np.load(os.path.join(golden_dir, "box_golden.npz"))
    lib = _hip.load()
    n = g["dec_enc"].shape[0]
    coords = torch.zeros((1, 4), dtype=torch.int32, device=DEV)
    sel = torch.arange(n, dtype=torch.int64, device=DEV)
    enc, anchors = _t(g["dec_enc"]), _t(g["dec_anchors"])   # named: a temporary would be freed before the launch
    for w in ("w1", "w2"):
        boxes = torch.empty((n, 7), dtype=torch.float32, device=DEV)
        check(lib.aabr_rpn_decode(ptr(coords), 0, ptr(sel), n, ptr(enc), 0, ptr(anchors), n,
                                  20.0, _hip.f32xn([4.0, 4.0, 2.0]), _hip.f32xn(g["weights_" + w].tolist()), 10000.0,
                                  ptr(boxes), stream()))
        # device division / sqrt are IEEE; floorf(x/pi + .5) is exact -> tolerance only for fma contraction
        np.testing.assert_allclose(boxes.cpu().numpy(), g["dec_" + w], rtol=1e-6, atol=1e-6)


def test_boxes_iou_3d_and_nms_vs_reference_torch_golden(golden_dir):
    from utils3d.rotate_nms_3d_torch import boxes_iou_3d
    from second.pytorch.core.box_torch_ops import rotate_nms_3d
    g = np.load(os.path.join(golden_dir, "box_golden.npz"))
    for flag in ("rpn_label_generation", "roi_label_generation", "eval", "rpn_post"):
        a = g["iou3d_%s_aug" % flag]
        aug = None if flag == "rpn_post" else {"target_Y": float(a[0]), "target_Z": float(a[1]),
                                                "anchor_Y": float(a[2]), "anchor_Z": float(a[3])}
        got = boxes_iou_3d(_t(g["iou3d_targets"]), _t(g["iou3d_anchors"]), aug, int(g["iou3d_%s_crit" % flag]),
                           only_xy=False, flag=flag).cpu().numpy()
        np.testing.assert_allclose(got, g["iou3d_" + flag], atol=2e-5)
    for pre, post in ((100, 30), (2000, 1000)):
        k = rotate_nms_3d(_t(g["nms3d_boxes"]), _t(g["nms3d_scores"]), pre, post, 0.5, flag="rpn_post").cpu().numpy()
        np.testing.assert_array_equal(k, g["nms3d_keep_%d_%d" % (pre, post)])

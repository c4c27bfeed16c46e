"""Brick-major site order (Metadata_3(site_order="brick"), csrc/brick.hip) against the CPU oracle on a real MI355X.

The oracle numbers sites like the reference / the default path (input level: first-seen, IOLayersRules.h:86-91; strided
levels: insertion order); the brick path numbers them brick by brick.  Parity is therefore the one SURVEY.md 7 defines:
the same SET of sites per sample, and every per-site quantity (features, rule partners, gradients) equal once rows are
matched by their coordinates -- integer structures exactly, floating point within the tolerance written at each assert."""
import numpy as np
import pytest
import torch

import oracle_lib as O

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


def _input(scn, coords, feats, spatial, mode=4, order="brick"):
    layer = scn.InputLayer(3, list(spatial), mode=mode)
    layer.site_order = order
    f = _t(feats).requires_grad_(True)
    return layer([_t(coords), f]), f


def _ckey(c):
    """one int64 per (x, y, z, b) row"""
    c = np.asarray(c, np.int64)
    return ((c[:, 3] * 70000 + c[:, 0]) * 70000 + c[:, 1]) * 70000 + c[:, 2]


def _match(dev_coords, ref_coords):
    """ref row of every device row (the two lists must hold the same sites)"""
    kd, kr = _ckey(dev_coords), _ckey(ref_coords)
    assert len(kd) == len(kr) and len(np.unique(kd)) == len(kd)
    order = np.argsort(kr)
    pos = np.searchsorted(kr[order], kd)
    assert (pos < len(kr)).all() and (kr[order][pos] == kd).all(), "site sets differ"
    return order[pos]


def _brick_major_key(c):
    """the order csrc/geom.h defines: batch, super-brick (x, y, z), brick in super-brick, cell in brick"""
    c = np.asarray(c, np.int64)
    x, y, z, b = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    sb = ((b * 5000 + (x >> 4)) * 5000 + (y >> 4)) * 5000 + (z >> 4)
    bit = (((x >> 2) & 3) << 4) | (((y >> 2) & 3) << 2) | ((z >> 2) & 3)
    cell = ((x & 3) << 4) | ((y & 3) << 2) | (z & 3)
    return (sb * 64 + bit) * 64 + cell


def _assert_brick_major(coords):
    k = _brick_major_key(coords)
    assert (np.diff(k) > 0).all(), "rows are not in brick-major order"


# ------------------------------------------------------------------------------------ input level
@pytest.fixture(params=["native", "renumbered"])
def scatter_form(request):
    """the two ways to a brick-ordered input level: the scatter straight into the brick grid (default), or the hash scatter
    followed by a renumbering of its first-seen sites (SCN.brick_scatter = False)"""
    from sparseconvnet import SCN
    old = SCN.brick_scatter
    SCN.brick_scatter = request.param == "native"
    yield request.param
    SCN.brick_scatter = old


def _rows(x, ref_coords):
    """(r, inv): device row i holds the oracle's row r[i]; inv = the other way"""
    r = _match(x.get_spatial_locations().numpy(), ref_coords)
    inv = np.empty_like(r)
    inv[r] = np.arange(len(r))
    return r, inv


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_brick_input_layer_is_a_row_permutation_of_the_oracle(mode, scatter_form):
    scn = _scn()
    rng = np.random.default_rng(110 + mode)
    coords, feats = _rand_scene(rng, 6000, (40, 23, 9), 3, 7)
    x, f = _input(scn, coords, feats, (64, 32, 16), mode)
    ref = O.input_layer(coords, feats, mode)
    md = x.metadata
    assert md.site_order == "brick" and md.grids[tuple(x.spatial_size.tolist())].brick is not None
    assert md.input["V"] == ref["V"]
    loc = md.getSpatialLocations(x.spatial_size).numpy()
    _assert_brick_major(loc)
    r = _match(loc, ref["coords"])                      # device row i holds the oracle's row r[i]
    inv = np.empty_like(r)
    inv[r] = np.arange(len(r))
    if scatter_form == "renumbered":                    # (that form keeps the permutation it applied)
        np.testing.assert_array_equal(md.input["old_of_new"].cpu().numpy(), r)
        np.testing.assert_array_equal(md.input["new_of_old"].cpu().numpy(), inv)
    else:
        from sparseconvnet import SCN
        assert md.grids[tuple(x.spatial_size.tolist())].keys is None and SCN.scatter_stats["brick"] > 0     # no hash table
    # the point -> site map, renumbered
    np.testing.assert_array_equal(md.input["point_site"].cpu().numpy(), inv[ref["point_voxel"]])
    # reference-format rule table, row-permuted
    hdr, rules = md.inputLayerRuleBook()
    assert hdr == [mode, ref["max_active"], 6000, ref["V"]]
    np.testing.assert_array_equal(rules.cpu().numpy(), ref["rules"][r])
    # features: the same operations per site -> bit-exact
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"][r])
    g = rng.standard_normal(ref["out"].shape).astype(np.float32)       # oracle row order
    x.features.backward(_t(g[r]))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), O.input_layer_bwd(ref, g))


def test_brick_input_edge_cases(scatter_form):
    scn = _scn()
    from sparseconvnet import SCN
    # single point, all points in one voxel, a site at the far corner of the layer, 3-column coordinates
    for coords, spatial in ((np.array([[3, 4, 5]], np.int64), (16, 16, 16)),
                            (np.tile(np.array([[7, 7, 7]], np.int64), (1000, 1)), (16, 16, 16)),
                            (np.array([[0, 0, 0, 0], [4095, 4095, 511, 1], [0, 0, 0, 0], [4095, 4095, 511, 0]], np.int64),
                             (4096, 4096, 512)),
                            (np.array([[15, 15, 15, 0], [16, 16, 16, 0], [16, 15, 15, 0], [3, 4, 5, 2]], np.int64),
                             (32, 32, 32))):
        feats = np.arange(coords.shape[0] * 2, dtype=np.float32).reshape(-1, 2)
        x, _ = _input(scn, coords, feats, spatial, 4)
        ref = O.input_layer(coords, feats, 4)
        loc = x.get_spatial_locations().numpy()
        if coords.shape[1] == 3:
            assert (loc[:, 3] == 0).all()
        r = _match(loc, np.concatenate([ref["coords"][:, :3], ref["coords"][:, 3:4] if ref["coords"].shape[1] == 4
                                         else np.zeros((ref["V"], 1), np.int64)], 1))
        _assert_brick_major(loc)
        np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"][r])
        tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
        assert sum(tb.rule_counts()) == O.submanifold_rules(ref["coords"], [3, 3, 3]).total
    # empty input
    x, _ = _input(scn, np.zeros((0, 4), np.int64), np.zeros((0, 3), np.float32), (8, 8, 8), 4)
    assert tuple(x.features.shape) == (0, 3) and x.get_spatial_locations().shape == (0, 4)
    tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    assert tb.V_out == 0
    # a few sites over a huge extent: the scene keeps its hash grids (first-seen order), loudly counted
    n0 = SCN.brick_stats["declined"]
    c = np.array([[0, 0, 0], [65534, 65534, 65534], [0, 0, 0]], np.int64)
    x, _ = _input(scn, c, np.ones((3, 2), np.float32), (70000, 70000, 70000), 4)
    assert SCN.brick_stats["declined"] == n0 + 1 and x.metadata.site_order == "first_seen"
    np.testing.assert_array_equal(x.get_spatial_locations().numpy()[:, :3], O.input_layer(c, np.ones((3, 2), np.float32), 4)["coords"][:, :3])


# ------------------------------------------------------------------------------------ rule books
def _table_of(rb, V):
    """oracle rule book -> gather table [vol, V] (input row per output row, -1)"""
    t = np.full((rb.vol, V), -1, np.int64)
    for k in range(rb.vol):
        p = rb.pairs(k)
        if len(p):
            t[k, p[:, 1]] = p[:, 0]
    return t


def test_brick_submanifold_tables_match_oracle_under_the_permutation():
    scn = _scn()
    rng = np.random.default_rng(120)
    coords, feats = _rand_scene(rng, 9000, (50, 37, 11), 3, 3)
    x, _ = _input(scn, coords, feats, (64, 64, 16), 3)
    ref_il = O.input_layer(coords, feats, 3)
    r, inv = _rows(x, ref_il["coords"])
    for fs in ([3, 3, 3], [1, 1, 1], [3, 1, 5]):
        tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor(fs))
        rb = O.submanifold_rules(ref_il["coords"], fs)
        np.testing.assert_array_equal(np.array(tb.rule_counts()), rb.counts)
        want = _table_of(rb, ref_il["V"])[:, r]                        # oracle partners of the device's rows ...
        want = np.where(want >= 0, inv[np.maximum(want, 0)], -1)       # ... in the device's numbering
        np.testing.assert_array_equal(tb.out.table.cpu().numpy(), want)


@pytest.mark.parametrize("fs,st", [([2, 2, 2], [2, 2, 2]), ([3, 3, 3], [2, 2, 2]), ([1, 1, 8], [1, 1, 1]),
                                   ([4, 4, 4], [2, 2, 2]), ([1, 1, 3], [1, 1, 1])])
def test_brick_strided_levels_and_tables_match_oracle(fs, st):
    scn = _scn()
    rng = np.random.default_rng(121)
    size = np.array([37, 35, 8]) if fs != [1, 1, 8] else np.array([44, 21, 8])
    if fs == [2, 2, 2]:
        size = np.array([48, 36, 8])
    if fs == [4, 4, 4]:
        size = np.array([38, 38, 10])
    coords, feats = _rand_scene(rng, 7000, tuple(size), 3, 3)
    x, _ = _input(scn, coords, feats, tuple(size), 3)
    ref_il = O.input_layer(coords, feats, 3)
    osz = (size - np.array(fs)) // np.array(st) + 1
    tb = x.metadata.getRuleBook(x.spatial_size, torch.LongTensor(osz), torch.LongTensor(fs), torch.LongTensor(st))
    rb, oc = O.convolution_rules(ref_il["coords"], fs, st, osz)
    loc_out = x.metadata.getSpatialLocations(torch.LongTensor(osz)).numpy()
    _assert_brick_major(loc_out)
    ro = _match(loc_out, oc)                                            # device output row -> oracle output row
    ri, _ = _rows(x, ref_il["coords"])                                  # device input row -> oracle input row
    np.testing.assert_array_equal(np.array(tb.rule_counts()), rb.counts)
    t_out = tb.out.table.cpu().numpy()                                  # [vol, V_out]: device input row per output row
    t_in = tb.inn.table.cpu().numpy()                                   # [vol, V_in]: device output row per input row
    for k in range(rb.vol):
        want = {(int(a), int(b)) for a, b in rb.pairs(k)}               # (oracle in, oracle out)
        o = np.nonzero(t_out[k] >= 0)[0]
        assert {(int(ri[t_out[k, j]]), int(ro[j])) for j in o} == want
        u = np.nonzero(t_in[k] >= 0)[0]
        assert {(int(ri[j]), int(ro[t_in[k, j]])) for j in u} == want
    # sample offsets of the new level (SparseGrid::ctr): rows are batch-contiguous in batch order
    g = x.metadata.grids[tuple(int(v) for v in osz)]
    assert (np.diff(loc_out[:, 3]) >= 0).all()
    if g.sample_off is not None:
        cnt = np.bincount(loc_out[:, 3], minlength=3)
        assert g.sample_counts(3) == cnt.tolist()


def test_brick_second_rule_book_onto_an_existing_output_level():
    """Two strided rule books with different filters onto ONE output size (the reference serves the second from the grid
    the first one built, Metadata.cpp:484-510): under site_order='brick' the existing level is used as it is -- same rows,
    no renumbering under features that already live on it, no fall-through to the hash builders (advisor, round 5) -- and
    the second book's tables are the oracle's rules under the two levels' permutations."""
    scn = _scn()
    rng = np.random.default_rng(123)
    size = np.array([48, 36, 8])
    coords, feats = _rand_scene(rng, 6000, tuple(size), 2, 3)
    x, _ = _input(scn, coords, feats, tuple(size), 3)
    md = x.metadata
    ref_il = O.input_layer(coords, feats, 3)
    osz = size // 2
    L = lambda v: torch.LongTensor(list(int(i) for i in v))
    tb1 = md.getRuleBook(x.spatial_size, L(osz), L([2, 2, 2]), L([2, 2, 2]))
    loc1 = md.getSpatialLocations(L(osz)).numpy().copy()
    g1 = md.grids[tuple(int(v) for v in osz)]
    tb2 = md.getRuleBook(x.spatial_size, L(osz), L([1, 1, 1]), L([2, 2, 2]))       # even sites only: a subset of the level
    assert md.grids[tuple(int(v) for v in osz)] is g1 and tb2 is not tb1
    np.testing.assert_array_equal(md.getSpatialLocations(L(osz)).numpy(), loc1)
    _, oc1 = O.convolution_rules(ref_il["coords"], [2, 2, 2], [2, 2, 2], osz)
    rb2, oc2 = O.convolution_rules(ref_il["coords"], [1, 1, 1], [2, 2, 2], osz)
    ro = _match(loc1, oc1)                                              # device output row -> oracle row of book 1's level
    ri, _ = _rows(x, ref_il["coords"])
    t_out = tb2.out.table.cpu().numpy()
    t_in = tb2.inn.table.cpu().numpy()
    assert rb2.vol == 1 and 0 < rb2.counts[0] < ref_il["V"]
    # oracle pairs of book 2 are (input row, row in ITS OWN output list oc2): compare through coordinates
    key = lambda c: tuple(int(v) for v in c)
    want = {(int(a), key(oc2[int(b)])) for a, b in rb2.pairs(0)}
    o = np.nonzero(t_out[0] >= 0)[0]
    assert {(int(ri[t_out[0, j]]), key(oc1[ro[j]])) for j in o} == want
    u = np.nonzero(t_in[0] >= 0)[0]
    assert {(int(ri[j]), key(oc1[ro[t_in[0, j]]])) for j in u} == want
    assert list(tb2.rule_counts()) == [int(rb2.counts[0])]


def test_brick_pyramid_one_read_and_chain_of_levels():
    """a chain of non-overlapping levels + a z-collapse level, built level from level with device-side counts"""
    scn = _scn()
    rng = np.random.default_rng(122)
    coords, feats = _rand_scene(rng, 20000, (200, 150, 30), 2, 3)
    x, _ = _input(scn, coords, feats, (256, 256, 32), 3)
    md = x.metadata
    ref_il = O.input_layer(coords, feats, 3)
    specs, sz, cur = [], (256, 256, 32), ref_il["coords"]
    want = {}
    for _ in range(4):
        osz = tuple(s // 2 for s in sz)
        specs.append((osz, sz, (2, 2, 2), (2, 2, 2)))
        _, cur = O.convolution_rules(cur, [2, 2, 2], [2, 2, 2], list(osz))
        want[osz] = cur
        sz = osz
    specs.append(((sz[0], sz[1], 1), sz, (1, 1, sz[2]), (1, 1, sz[2])))
    _, c2d = O.convolution_rules(cur, [1, 1, sz[2]], [1, 1, 1], [sz[0], sz[1], 1])
    want[(sz[0], sz[1], 1)] = c2d
    import _hip
    md.buildBrickPyramid(specs)
    for osz, w in want.items():
        loc = md.getSpatialLocations(torch.LongTensor(list(osz))).numpy()
        _match(loc, w)
        _assert_brick_major(loc)
        assert md.grids[osz].sample_counts(2) == np.bincount(w[:, 3], minlength=2).tolist()


# ------------------------------------------------------------------------------------ convolutions over brick grids
@pytest.mark.parametrize("nIn,nOut", [(9, 32), (32, 32), (64, 64), (128, 128)])
def test_brick_submanifold_conv_forward_backward(nIn, nOut):
    scn = _scn()
    rng = np.random.default_rng(nIn * 1000 + nOut + 7)
    coords, feats = _rand_scene(rng, 4000, (30, 22, 6), 2, nIn)
    x, f = _input(scn, coords, feats, (32, 32, 8), 3)
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, False).to(DEV)
    y = conv(x)
    ref_il = O.input_layer(coords, feats, 3)
    r, _ = _rows(x, ref_il["coords"])
    rb = O.submanifold_rules(ref_il["coords"], [3, 3, 3])
    W = conv.weight.detach().cpu().numpy().reshape(27, nIn, nOut)
    want, _ = O.conv_fwd(ref_il["out"], W, rb, ref_il["V"])
    got = y.features.detach().cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want[r]).max() <= 1e-4 * scale + 2e-6 * scale * nIn        # as test_gpu_parity's conv test
    g = rng.standard_normal(want.shape).astype(np.float32)
    y.features.backward(_t(g[r]))
    d_in, dW, _ = O.conv_bwd(ref_il["out"], g, W, rb)
    gW = conv.weight.grad.cpu().numpy().reshape(dW.shape)
    assert np.abs(gW - dW).max() <= 2e-4 * np.abs(dW).max()
    want_pts = O.input_layer_bwd(ref_il, d_in.astype(np.float32))
    assert np.abs(f.grad.cpu().numpy() - want_pts).max() <= 2e-4 * np.abs(want_pts).max()


def test_brick_fpn_net_equals_first_seen_fpn_net_site_by_site():
    """the whole backbone (compiled graph) in brick order against the SAME network in the reference's order (which
    tests/test_gpu_fpn.py pins to the oracle): returned maps, input gradient and parameter gradients"""
    scn = _scn()
    import synth_scenes as S
    from test_cabi_and_host import default_fpn
    torch.manual_seed(3)
    net = default_fpn().to(DEV)
    net.compiled_graph = True
    locs, feats = S.make_batch(2, 6000, 40, 50)
    l, f0 = _t(locs), _t(feats)

    def run(order):
        net.set_site_order(order)
        for p in net.parameters():
            p.grad = None
        f = f0.clone().requires_grad_(True)
        rpn, roi = net([l, f])
        maps = rpn + roi
        loss = sum((m.features * (1.0 + 0.01 * i)).square().mean() for i, m in enumerate(maps))
        loss.backward()
        return ([(m.get_spatial_locations().numpy(), m.features.detach().cpu().numpy()) for m in maps],
                f.grad.cpu().numpy(), [p.grad.cpu().numpy().copy() if p.grad is not None else None for p in net.parameters()])

    a_maps, a_dx, a_gp = run("first_seen")
    b_maps, b_dx, b_gp = run("brick")
    for (la, fa), (lb, fb) in zip(a_maps, b_maps):
        r = _match(lb, la)
        _assert_brick_major(lb)
        assert np.abs(fb - fa[r]).max() <= 2e-4 * np.abs(fa).max()
    # gradients: ~100 layers, summation orders differ (dW chunks, BatchNorm statistics) and single ReLU masks flip at
    # rounding distance of 0 -- a flip is a discrete change, so the bound is on the relative L2 error of each tensor
    # (the per-operator tests above hold the tight, flip-free tolerances)
    def rel_l2(a, b):
        return float(np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(a.astype(np.float64)) + 1e-30))
    assert rel_l2(a_dx, b_dx) <= 1e-2
    worst = 0.0
    for ga, gb in zip(a_gp, b_gp):
        assert (ga is None) == (gb is None)
        if ga is not None:
            worst = max(worst, rel_l2(ga, gb))
    assert worst <= 1e-2, worst


def test_prepare_in_two_halves_gives_the_same_pass():
    """FPN_Net.prepare_begin / prepare_end (the level pyramid enqueued, its counts collected later, other launches in
    between) against a pass without any prefetch: same maps, bit for bit"""
    scn = _scn()
    import synth_scenes as S
    from test_cabi_and_host import default_fpn
    torch.manual_seed(4)
    net = default_fpn().to(DEV)
    net.compiled_graph = True
    net.set_site_order("brick")
    locs, feats = S.make_batch(2, 6000, 50, 50)
    l, f = _t(locs), _t(feats)
    with torch.no_grad():
        a_rpn, a_roi = net([l, f])
        want = [(m.get_spatial_locations().numpy(), m.features.clone()) for m in a_rpn + a_roi]
        side = torch.cuda.Stream()
        net.prepare_begin([l, f], side)
        filler = torch.randn(1024, 1024, device=DEV) @ torch.randn(1024, 1024, device=DEV)    # other work in between
        md = net._preparing[0]
        assert md.__dict__.get("_pyramid_pending") is not None and len(md.grids) == 1       # levels enqueued, not collected
        net.prepare_end()
        assert len(md.grids) > 9 and md.__dict__.get("_pyramid_pending") is None
        b_rpn, b_roi = net([l, f])
        assert b_rpn[0].metadata is md                                                      # the prepared geometry was used
    for (lw, fw), m in zip(want, b_rpn + b_roi):
        np.testing.assert_array_equal(m.get_spatial_locations().numpy(), lw)
        assert torch.equal(m.features, fw)
    del filler

"""The three forms of the voxel scatter's insert (csrc/voxel_scatter.hip; reference: SCN/Metadata/IOLayersRules.h:18-125,
SCN/CPU/IOLayers.cpp:11-47) against the oracle: generic (two device atomics per point), packed (one) and LDS-binned
(none on the table) must give the SAME bits -- site list in first-seen order, point -> site map, rule table, features,
input gradient -- and the grid they leave behind must serve the rule-book builders identically.  Points that do not
fit the packed word, and hash blocks that overflow, fall back to the generic form."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth_scenes as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture
def scn_mod():
    import sparseconvnet as scn
    from sparseconvnet import SCN
    keep = (SCN.scatter_variant, SCN.SCATTER_MIN_POINTS)
    yield scn, SCN
    SCN.scatter_variant, SCN.SCATTER_MIN_POINTS = keep


def _run(scn, coords, feats, spatial, mode=4):
    layer = scn.InputLayer(3, list(spatial), mode=mode)
    f = _t(feats).requires_grad_(True)
    return layer([_t(coords), f]), f


def _check(scn, SCN, coords, feats, spatial, mode, variant, expect_variant=None, expect_redo=False):
    SCN.scatter_variant = variant
    before = dict(SCN.scatter_stats)
    x, f = _run(scn, coords, feats, spatial, mode)
    ran = [k for k in ("variant0", "variant1", "variant2") if SCN.scatter_stats[k] != before[k]]
    assert ran == ["variant%d" % (variant if expect_variant is None else expect_variant)], ran
    assert (SCN.scatter_stats["redone"] != before["redone"]) == expect_redo
    ref = O.input_layer(coords, feats, mode)
    md = x.metadata
    assert md.input["V"] == ref["V"]
    np.testing.assert_array_equal(md.getSpatialLocations(x.spatial_size).numpy(), ref["coords"])
    np.testing.assert_array_equal(md.input["point_site"].cpu().numpy(), ref["point_voxel"])
    hdr, rules = md.inputLayerRuleBook()
    assert hdr == [mode, ref["max_active"], coords.shape[0], ref["V"]]
    np.testing.assert_array_equal(rules.cpu().numpy(), ref["rules"])
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"])
    g = np.random.default_rng(1).standard_normal(ref["out"].shape).astype(np.float32)
    x.features.backward(_t(g))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), O.input_layer_bwd(ref, g))
    # the grid left behind: the submanifold rule book built by probing it == the oracle's
    tb = md.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
    rb = O.submanifold_rules(ref["coords"], [3, 3, 3])
    assert tb.out.rule_counts() == rb.counts.tolist()
    return x


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("mode", [1, 4])
def test_scatter_forms_agree_with_oracle_on_a_scene(scn_mod, variant, mode):
    scn, SCN = scn_mod
    SCN.SCATTER_MIN_POINTS = 1000
    locs, feats = S.make_batch(3, 30000, 17, 50)              # ~90k points, 3 samples, 2 cm
    _check(scn, SCN, locs, feats, S.FULL_SCALE, mode, variant)


@pytest.mark.parametrize("variant", [1, 2])
def test_scatter_heavy_duplication_and_small_grids(scn_mod, variant):
    scn, SCN = scn_mod
    SCN.SCATTER_MIN_POINTS = 1
    rng = np.random.default_rng(5)
    # ~3 points per voxel in a 12 x 10 x 6 box, 3 samples
    coords = np.stack([rng.integers(0, s, 20000) for s in (12, 10, 6)] + [np.sort(rng.integers(0, 3, 20000))], 1)
    feats = rng.standard_normal((20000, 7)).astype(np.float32)
    _check(scn, SCN, coords.astype(np.int64), feats, (16, 16, 8), 4, variant)
    # a tiny input: cap = 64 < one hash block -> the binned form steps down to the packed one
    small = coords[:20].astype(np.int64)
    small[:, 3] = 0
    _check(scn, SCN, small, feats[:20], (16, 16, 8), 3, variant, expect_variant=1)
    # quantiser sentinels (-1, -1, -1) are skipped silently by every form
    c2 = coords.astype(np.int64).copy()
    c2[::7, :3] = -1
    x = None
    SCN.scatter_variant = variant
    x, _ = _run(scn, c2, feats, (16, 16, 8), 4)
    keep = (c2[:, 0] >= 0)
    ref = O.input_layer(c2[keep], feats[keep], 4)
    assert x.metadata.input["V"] == ref["V"]
    np.testing.assert_array_equal(x.get_spatial_locations().numpy(), ref["coords"])
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"])
    ps = x.metadata.input["point_site"].cpu().numpy()
    assert (ps[~keep] == -1).all() and (ps[keep] == ref["point_voxel"]).all()


@pytest.mark.parametrize("variant", [1, 2])
def test_scatter_fallbacks(scn_mod, variant):
    """what does not fit the 64-bit word runs through the generic form: a coordinate beyond the layer's spatial size,
    a batch index beyond the bits left, every point in ONE voxel (its hash block's record region overflows: binned
    form only), a spatial size whose fields leave no room; below the size threshold the generic form is chosen up
    front; coordinates beyond 65534 are still rejected loudly"""
    scn, SCN = scn_mod
    import _hip
    SCN.SCATTER_MIN_POINTS = 1000
    locs, feats = S.make_batch(2, 20000, 3, 50)
    n = locs.shape[0]
    far = locs.copy()
    far[5, 0] = 5000                                            # >= 4096: beyond the x field
    _check(scn, SCN, far, feats, S.FULL_SCALE, 4, variant, expect_redo=True)
    bat = locs.copy()
    bat[-3:, 3] = 40000                                         # batch field: 64 - 33 - 16 = 15 bits -> < 32767
    _check(scn, SCN, bat, feats, S.FULL_SCALE, 4, variant, expect_redo=True)
    one = np.tile(np.array([[7, 7, 7, 0]], np.int64), (6000, 1))
    f1 = np.arange(12000, dtype=np.float32).reshape(-1, 2)
    _check(scn, SCN, one, f1, S.FULL_SCALE, 4, variant, expect_redo=(variant == 2))
    # spatial size 65535^3 + 40k points: 48 + 16 bits, nothing left for the batch index -> generic up front
    _check(scn, SCN, locs, feats, (65535, 65535, 65535), 4, variant, expect_variant=0)
    SCN.SCATTER_MIN_POINTS = n + 1
    _check(scn, SCN, locs, feats, S.FULL_SCALE, 4, variant, expect_variant=0)
    SCN.SCATTER_MIN_POINTS = 1000
    bad = locs.copy()
    bad[9, 2] = 70000
    with pytest.raises(_hip.AabrError):
        _run(scn, bad, feats, S.FULL_SCALE, 4)
    SCN.scatter_variant = variant
    bad[9, 2] = -3
    with pytest.raises(_hip.AabrError):
        _run(scn, bad, feats, S.FULL_SCALE, 4)


def test_scatter_forms_bitwise_equal_at_1p5M(scn_mod):
    """BASELINE configs[4] size: the three forms leave the same site list, point map and features"""
    scn, SCN = scn_mod
    locs, feats = S.make_batch(1, 1500000, 0, 50)
    l, f = _t(locs), _t(feats)
    outs = []
    for variant in (0, 1, 2):
        SCN.scatter_variant = variant
        x = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)([l, f])
        outs.append((x.metadata.input["V"], x.metadata.getSpatialLocationsDevice(x.spatial_size).clone(),
                     x.metadata.input["point_site"].clone(), x.features.clone()))
    assert outs[0][0] > 800000
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        for a, b in zip(o[1:], outs[0][1:]):
            assert torch.equal(a, b)

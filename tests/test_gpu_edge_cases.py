"""Boundary shapes of the HIP path against the oracle: tile / block / chunk edges, ragged batches,
empty samples, mode 0, score ties and oversize NMS inputs."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth_scenes as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scn():
    import sparseconvnet as scn
    return scn


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _rand_scene(rng, n, size, batch, C):
    coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, batch, n))], 1)
    return coords.astype(np.int64), rng.standard_normal((n, C)).astype(np.float32)


def _unique_sites(rng, V, size=(9, 9, 9), batch=1):
    """exactly V distinct sites (dense enough that most have neighbours), batch-sorted"""
    cells = np.stack(np.meshgrid(*[np.arange(s) for s in size], indexing="ij"), -1).reshape(-1, 3)
    out = []
    per = [V // batch + (1 if b < V % batch else 0) for b in range(batch)]
    for b in range(batch):
        idx = rng.choice(len(cells), per[b], replace=False)
        out.append(np.concatenate([cells[idx], np.full((per[b], 1), b)], 1))
    return np.concatenate(out, 0).astype(np.int64)


@pytest.mark.parametrize("V", [1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 255, 256, 257, 700])
def test_conv_forward_backward_at_tile_and_chunk_edges(V):
    scn = _scn()
    rng = np.random.default_rng(V)
    coords = _unique_sites(rng, V)
    nIn, nOut = 32, 48
    feats = rng.standard_normal((V, nIn)).astype(np.float32)
    layer = scn.InputLayer(3, [16, 16, 16], mode=4)
    f = _t(feats).requires_grad_(True)
    x = layer([_t(coords), f])
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, True).to(DEV)
    conv.bias.data.normal_()
    y = conv(x)
    il = O.input_layer(coords, feats, 4)
    assert il["V"] == V
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = conv.weight.detach().cpu().numpy().reshape(27, nIn, nOut)
    ref, _ = O.conv_fwd(il["out"], W, rb, V, conv.bias.detach().cpu().numpy())
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, db = O.conv_bwd(il["out"], g, W, rb, want_bias=True)
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.input_layer_bwd(il, d_in), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(dW.shape), dW, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4)


def test_ragged_batch_with_empty_sample_and_strided_chain():
    """samples 0, 2, 3 present, sample 1 empty; two stride-2 convolutions and the transposes back"""
    scn = _scn()
    rng = np.random.default_rng(7)
    parts = []
    for b, n in ((0, 300), (2, 5), (3, 900)):
        c = np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n), rng.integers(0, 8, n), np.full(n, b)], 1)
        parts.append(c)
    coords = np.concatenate(parts, 0).astype(np.int64)
    feats = rng.standard_normal((coords.shape[0], 16)).astype(np.float32)
    x = scn.InputLayer(3, [16, 16, 8], mode=3)([_t(coords), _t(feats), 4])
    c1 = scn.Convolution(3, 16, 16, 2, 2, False).to(DEV)
    c2 = scn.Convolution(3, 16, 16, 2, 2, False).to(DEV)
    d2 = scn.Deconvolution(3, 16, 16, 2, 2, False).to(DEV)
    d1 = scn.Deconvolution(3, 16, 16, 2, 2, False).to(DEV)
    with torch.no_grad():
        y1 = c1(x)
        y2 = c2(y1)
        z = d1(d2(y2))
    il = O.input_layer(coords, feats, 3)
    rb1, oc1 = O.convolution_rules(il["coords"], [2, 2, 2], [2, 2, 2], [8, 8, 4])
    rb2, oc2 = O.convolution_rules(oc1, [2, 2, 2], [2, 2, 2], [4, 4, 2])
    np.testing.assert_array_equal(y1.get_spatial_locations().numpy(), oc1)
    np.testing.assert_array_equal(y2.get_spatial_locations().numpy(), oc2)
    assert set(np.unique(oc2[:, 3])) == {0, 2, 3} and (np.diff(oc2[:, 3]) >= 0).all()
    P = lambda m: m.weight.detach().cpu().numpy().reshape(8, 16, 16)
    r1, _ = O.conv_fwd(il["out"], P(c1), rb1, oc1.shape[0])
    r2, _ = O.conv_fwd(r1, P(c2), rb2, oc2.shape[0])
    u2, _ = O.conv_fwd(r2, P(d2), rb2, oc1.shape[0], in_col=1)
    u1, _ = O.conv_fwd(u2, P(d1), rb1, il["V"], in_col=1)
    np.testing.assert_allclose(z.features.cpu().numpy(), u1, rtol=1e-4, atol=1e-4 * np.abs(u1).max())
    np.testing.assert_array_equal(z.get_spatial_locations().numpy(), il["coords"])


def test_input_layer_mode0_and_duplicate_rejection():
    scn = _scn()
    import _hip
    rng = np.random.default_rng(8)
    coords = _unique_sites(rng, 300, (8, 8, 8), 2)
    feats = rng.standard_normal((300, 4)).astype(np.float32)
    x = scn.InputLayer(3, [8, 8, 8], mode=0)([_t(coords), _t(feats)])
    np.testing.assert_array_equal(x.features.cpu().numpy(), feats)  # unique input: identity
    np.testing.assert_array_equal(x.get_spatial_locations().numpy(), coords)
    dup = np.concatenate([coords, coords[:1]], 0)
    with pytest.raises(_hip.AabrError):
        scn.InputLayer(3, [8, 8, 8], mode=0)([_t(dup), _t(np.zeros((301, 4), np.float32))])


def test_zcollapse_conv_vol32_matches_oracle():
    """convs_pro2d geometry: filter [1,1,32] stride 1 (32 offsets, the largest volume FPN_Net uses)"""
    scn = _scn()
    rng = np.random.default_rng(9)
    n = 4000
    coords = np.stack([rng.integers(0, 24, n), rng.integers(0, 24, n), rng.integers(0, 32, n),
                       np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    feats = rng.standard_normal((n, 128)).astype(np.float32)
    f = _t(feats).requires_grad_(True)
    x = scn.InputLayer(3, [24, 24, 32], mode=4)([_t(coords), f])
    conv = scn.Convolution(3, 128, 128, [1, 1, 32], [1, 1, 1], False).to(DEV)
    y = conv(x)
    assert y.spatial_size.tolist() == [24, 24, 1]
    il = O.input_layer(coords, feats, 4)
    rb, oc = O.convolution_rules(il["coords"], [1, 1, 32], [1, 1, 1], [24, 24, 1])
    W = conv.weight.detach().cpu().numpy().reshape(32, 128, 128)
    ref, _ = O.conv_fwd(il["out"], W, rb, oc.shape[0])
    np.testing.assert_array_equal(y.get_spatial_locations().numpy(), oc)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, _ = O.conv_bwd(il["out"], g, W, rb)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(dW.shape), dW, rtol=1e-4,
                               atol=2e-5 * np.abs(dW).max())
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.input_layer_bwd(il, d_in), rtol=1e-4,
                               atol=2e-5 * np.abs(d_in).max())


def test_nms_more_boxes_than_pre_max_and_score_ties():
    from second.pytorch.core.box_torch_ops import rotate_nms_3d
    b7, sc = S.make_nms_boxes(3000, 21)
    got = rotate_nms_3d(_t(b7), _t(sc), pre_max_size=2000, post_max_size=1000, iou_threshold=0.5, flag="rpn_post")
    want = O.rotate_nms_3d(b7, sc, 2000, 1000, 0.5)
    assert set(got.cpu().numpy().tolist()) == set(want.tolist())
    # tied scores: survivors are mutually non-overlapping and every dropped box overlaps a kept one
    sc_t = np.round(sc * 4) / 4
    keep = rotate_nms_3d(_t(b7[:500]), _t(sc_t[:500]), None, None, 0.5).cpu().numpy()
    iou = O.boxes_iou_3d(b7[:500], b7[:500])
    sub = iou[np.ix_(keep, keep)] - np.eye(len(keep))
    assert (sub < 0.5).all()
    dropped = np.setdiff1d(np.arange(500), keep)
    assert (iou[np.ix_(dropped, keep)].max(1) >= 0.5).all()


def test_device_quantisation_pipeline_matches_host_rule():
    """A1: xyz (metres) -> locs/feats on the device vs the dataset's host rule
    (suncg_dataset.py:126-188: scale, shift to the minimum, drop outside FULL_SCALE, truncate),
    then through the InputLayer: dropped points are skipped via the coordinate sentinel."""
    scn = _scn()
    import input_pipeline
    full = (300, 200, 60)  # small FULL_SCALE so that some points fall outside and are dropped
    xyz_l, ex_l, ref_locs, ref_feats = [], [], [], []
    for b in range(2):
        xyz, rng = S.synth_points(6000, 30 + b)
        extra = rng.random((xyz.shape[0], 6)).astype(np.float32)
        a = xyz * 20.0
        a = a - a.min(0)
        keep = (a.min(1) >= 0) & (a < np.array(full)).all(1)
        assert 0 < keep.sum() < len(keep)
        ref_locs.append(np.concatenate([np.trunc(a[keep]).astype(np.int64), np.full((keep.sum(), 1), b)], 1))
        ref_feats.append(np.concatenate([(a[keep] / 20.0).astype(np.float32), extra[keep]], 1))
        xyz_l.append(_t(xyz))
        ex_l.append(_t(extra))
    locs, feats = input_pipeline.quantize_scenes(xyz_l, ex_l, 20.0, full)
    kept = (locs[:, 0] >= 0).cpu().numpy()
    np.testing.assert_array_equal(locs.cpu().numpy()[kept], np.concatenate(ref_locs, 0))
    np.testing.assert_array_equal(feats.cpu().numpy()[kept], np.concatenate(ref_feats, 0))
    assert (locs.cpu().numpy()[~kept, :3] == -1).all()
    f = feats.clone().requires_grad_(True)
    x = scn.InputLayer(3, list(full), mode=4)([locs, f])
    il = O.input_layer(np.concatenate(ref_locs, 0), np.concatenate(ref_feats, 0), 4)
    np.testing.assert_array_equal(x.get_spatial_locations().numpy(), il["coords"])
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), il["out"])
    g = np.random.default_rng(0).standard_normal(il["out"].shape).astype(np.float32)
    x.features.backward(_t(g))
    gin = f.grad.cpu().numpy()
    np.testing.assert_array_equal(gin[kept], O.input_layer_bwd(il, g))
    assert (gin[~kept] == 0).all()  # dropped points receive no gradient


def _np_decode(reg, anchors, weights, clip):
    """oracle/box_oracle.py (pinned by tests/golden/box_golden.npz against the reference's BoxCoder3D)"""
    import sys
    sys.path.insert(0, O.ORACLE_DIR)
    import box_oracle
    return box_oracle.decode_centroid_box(reg, anchors, weights, clip)


def test_rpn_glue_anchors_decode_nms_on_device():
    scn = _scn()
    import rpn_glue
    rng = np.random.default_rng(17)
    n = 3000
    coords = np.stack([rng.integers(0, 64, n), rng.integers(0, 64, n), rng.integers(0, 8, n),
                       np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    x = scn.InputLayer(3, [64, 64, 8], mode=3)([_t(coords), _t(np.zeros((n, 1), np.float32))])
    sites = x.get_spatial_locations().numpy()
    V, A = sites.shape[0], 4
    base = np.zeros((A, 7), np.float32)
    base[:, 3:6] = [0.2, 2.0, 2.6]
    base[:, 6] = [0.0, np.pi / 4, np.pi / 2, -np.pi / 4]
    stride = [4.0, 4.0, 2.0]
    weights = (1.0, 1.0, 1.0, 2.0, 2.0, 2.0, 1.5)
    obj = rng.standard_normal(V * A).astype(np.float32)
    reg = (rng.standard_normal((V * A, 7)) * 0.3).astype(np.float32)
    res = rpn_glue.rpn_proposals_single_map(x, _t(obj), _t(reg), torch.as_tensor(base), 20.0, stride, 500, 100,
                                            0.5, (0.3, 0.3), weights, 10000.0)
    # anchors: grid_anchors vs the reference formula in numpy float32
    anc_d = rpn_glue.grid_anchors(x.metadata.grids[(64, 64, 8)].coords, torch.as_tensor(base), 20.0, stride)
    cen = (sites[:, :3].astype(np.float32) / np.float32(20.0) * np.asarray(stride, np.float32)).astype(np.float32)
    anc = (np.concatenate([cen, np.zeros((V, 4), np.float32)], 1)[:, None, :] + base[None]).reshape(-1, 7)
    # (torch's device-side float division may differ from IEEE by an ulp; the fused decode kernel divides exactly)
    np.testing.assert_allclose(anc_d.cpu().numpy(), anc, rtol=3e-7, atol=1e-6)
    assert len(res) == 2
    s = 0
    for bi in range(2):
        e = s + int((sites[:, 3] == bi).sum())
        o = 1.0 / (1.0 + np.exp(-obj[s * A:e * A].astype(np.float64)))
        idx = np.argsort(-o, kind="stable")[:500]
        dec = _np_decode(reg[s * A:e * A][idx], anc[s * A:e * A][idx], weights, 10000.0)
        nb = dec.copy()
        nb[:, 3:5] = np.maximum(nb[:, 3:5], 0.3)
        nb[:, 5] = np.maximum(nb[:, 5], 0.3)
        iou = O.boxes_iou_3d(nb, nb)
        keep = O.nms_from_matrix(iou, np.arange(len(idx), dtype=np.int32), 0.5)[:100]
        boxes_d, score_d = res[bi]
        np.testing.assert_allclose(boxes_d.cpu().numpy(), dec[keep], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(score_d.cpu().numpy(), o[idx][keep], rtol=1e-5, atol=1e-6)
        s = e


def test_sparse_to_dense_and_rotated_roi_align_3d():
    """§8f rank 3: SparseToDense (SCN/CPU/SparseToDense.cpp) + `_C.roi_align_rotated_3d_*`
    (csrc/cuda/ROIAlignRotated3D_cuda.cu) through the reference's ROIAlignRotated3D module"""
    scn = _scn()
    from maskrcnn_benchmark.layers import ROIAlignRotated3D
    rng = np.random.default_rng(31)
    n = 2500
    coords = np.stack([rng.integers(0, 20, n), rng.integers(0, 14, n), rng.integers(0, 5, n),
                       np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    feats = rng.standard_normal((n, 6)).astype(np.float32)
    f = _t(feats).requires_grad_(True)
    x = scn.InputLayer(3, [24, 16, 6], mode=4)([_t(coords), f])
    il = O.input_layer(coords, feats, 4)
    dense = scn.SparseToDense(3, 6)(x)
    ref_dense = O.sparse_to_dense(il["coords"], il["out"], [24, 16, 6], 2)
    np.testing.assert_array_equal(dense.detach().cpu().numpy(), ref_dense)
    # rois: (batch, center_w, center_h, center_z, width, height, zsize, theta[deg]) incl. one partly outside
    rois = np.array([[0, 6.0, 9.0, 2.0, 5.0, 8.0, 3.0, 30.0], [1, 3.0, 4.0, 1.0, 4.0, 4.0, 2.0, -75.0],
                     [1, 13.0, 19.0, 4.5, 9.0, 6.0, 4.0, 10.0], [0, 1.0, 1.0, 0.5, 0.2, 0.3, 0.1, 0.0]], np.float32)
    layer = ROIAlignRotated3D((3, 4, 2), 1.0, 2)
    out = layer(x, _t(rois))
    mx = il["coords"].max(0) + 1
    crop = ref_dense[:, :, :mx[0], :mx[1], :mx[2]]
    want = O.roi_align_rot3d_fwd(crop, rois, 1.0, (3, 4, 2), 2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-5)
    g = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(_t(g))
    gd = O.roi_align_rot3d_bwd(g, rois, 1.0, (3, 4, 2), crop.shape, 2)
    full = np.zeros_like(ref_dense)
    full[:, :, :mx[0], :mx[1], :mx[2]] = gd
    d_sparse = O.sparse_to_dense_bwd(il["coords"], full, 6, [24, 16, 6])
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.input_layer_bwd(il, d_sparse), rtol=1e-4, atol=1e-5)
    # adaptive sampling grid (sampling_ratio <= 0) and a spatial scale
    out2 = ROIAlignRotated3D((2, 2, 2), 0.5, 0)(x, _t(rois * np.array([1, 2, 2, 2, 2, 2, 2, 1], np.float32)))
    want2 = O.roi_align_rot3d_fwd(crop, rois * np.array([1, 2, 2, 2, 2, 2, 2, 1], np.float32), 0.5, (2, 2, 2), 0)
    np.testing.assert_allclose(out2.detach().cpu().numpy(), want2, rtol=1e-4, atol=1e-5)


def test_conv_backward_without_input_gradient_and_prepacked_weights():
    """autograd contract of the three convolution modules: when the input does not require a gradient the
    input-gradient pass is skipped (grad is None) and dW is unchanged; when it does, the backward pass
    reuses the weight pack made at forward time -- same bits as a stand-alone pack"""
    scn = _scn()
    rng = np.random.default_rng(77)
    n = 3000
    coords = np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n), rng.integers(0, 8, n),
                       np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    feats = rng.standard_normal((n, 32)).astype(np.float32)
    mods = [scn.SubmanifoldConvolution(3, 32, 64, 3, False), scn.Convolution(3, 64, 32, [2, 2, 2], [2, 2, 2], False),
            scn.Deconvolution(3, 32, 32, [2, 2, 2], [2, 2, 2], False)]
    mods = [m.to(DEV) for m in mods]
    results = []
    for need in (True, False):
        f = _t(feats).requires_grad_(need)
        x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), f])
        y = x
        for m in mods:
            m.zero_grad()
            y = m(y)
        g = torch.Generator(device=DEV).manual_seed(3)
        y.features.backward(torch.randn(y.features.shape, device=DEV, generator=g))
        results.append(([m.weight.grad.clone() for m in mods], f.grad, y.features.detach().clone()))
    (dw_a, gin_a, out_a), (dw_b, gin_b, out_b) = results
    assert gin_a is not None and gin_b is None
    assert torch.equal(out_a, out_b)
    # layers 2 and 3 still propagate (their inputs come from a layer with parameters); only the first
    # layer's input-gradient pass disappears, and no weight gradient changes by a bit
    for a, b in zip(dw_a, dw_b):
        assert torch.equal(a, b)
    # the reference-named entry point without the holder packs on its own: identical result
    import sparseconvnet.SCN as SCN
    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(feats)])
    d_out = torch.randn(x.features.shape[0], 64, device=DEV)
    w = mods[0].weight.detach()
    d_in_plain, d_in_held = torch.empty(0, device=DEV), torch.empty(0, device=DEV)
    dW1, dW2 = torch.empty_like(w), torch.empty_like(w)
    empty = torch.empty(0, device=DEV)
    SCN.SubmanifoldConvolution_backward(x.spatial_size, mods[0].filter_size, x.metadata, x.features, d_in_plain,
                                        d_out, w, dW1, empty)
    holder = []
    out = torch.empty(0, device=DEV)
    SCN.SubmanifoldConvolution_updateOutput(x.spatial_size, mods[0].filter_size, x.metadata, x.features, out, w,
                                            empty, pack_t=holder)
    assert len(holder) == 1
    SCN.SubmanifoldConvolution_backward(x.spatial_size, mods[0].filter_size, x.metadata, x.features, d_in_held,
                                        d_out, w, dW2, empty, pack_t=holder)
    assert torch.equal(d_in_plain, d_in_held) and torch.equal(dW1, dW2)


@pytest.mark.parametrize("nIn,nOut,npts", [(128, 128, 30000), (96, 256, 30000), (32, 64, 400000), (64, 64, 400000),
                                           (96, 256, 160000)])
def test_large_rule_book_kernel_variants_match_oracle(nIn, nOut, npts):
    """The dispatch in aabr_conv_forward picks by shape AND size: the weight-prefetching 64-column kernel
    needs >= 512 (tile, slab) workgroups, the LDS-resident-weight kernel only loops over several tiles
    per wave above ~130k sites, adjacent block pairs share their weight fragments above 8192 workgroups
    (the 160k-point, 256-plane case: 2,300 tiles x 4 slabs).  Small parity scenes never reach those paths; these do (forward and the
    transposed input-gradient pass, against the oracle on the same rule book)."""
    scn = _scn()
    rng = np.random.default_rng(nIn + nOut)
    locs, _ = S.make_batch(1, npts, 21, 50 if npts > 100000 else 20)  # 2 cm voxels: ~0.9 sites per point
    feats = rng.standard_normal((locs.shape[0], 4)).astype(np.float32)
    x = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)([_t(locs), _t(feats)])
    V = x.features.shape[0]
    assert V > (130000 if npts > 100000 else 16000)
    il = O.input_layer(locs, feats, 4)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    fin = rng.standard_normal((V, nIn)).astype(np.float32)
    xin = scn.SparseConvNetTensor(_t(fin).requires_grad_(True), x.metadata, x.spatial_size)
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, False).to(DEV)
    y = conv(xin)
    W = conv.weight.detach().cpu().numpy().reshape(27, nIn, nOut)
    ref, _ = O.conv_fwd(fin, W, rb, V)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max() * nIn)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, _ = O.conv_bwd(fin, g, W, rb)
    np.testing.assert_allclose(xin.features.grad.cpu().numpy(), d_in, rtol=1e-4,
                               atol=2e-6 * np.abs(d_in).max() * nOut)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(27, nIn, nOut), dW, rtol=2e-4,
                               atol=2e-5 * np.abs(dW).max())


def test_randomised_conv_shapes_against_oracle():
    """seeded sweep over plane counts (aligned and not), site counts around the 64-row tile boundaries and
    all three layer types: whatever kernel variant the dispatch picks must agree with the oracle"""
    scn = _scn()
    rng = np.random.default_rng(2024)
    plane_choices = [1, 3, 9, 16, 31, 32, 33, 48, 64, 65, 96, 128, 160]
    for trial in range(14):
        nIn, nOut = int(rng.choice(plane_choices)), int(rng.choice(plane_choices))
        n = int(rng.choice([1, 2, 63, 64, 65, 200, 700, 1500]))
        size = (9, 7, 5)
        coords = np.stack([rng.integers(0, s, n) for s in size] + [np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
        feats = rng.standard_normal((n, nIn)).astype(np.float32)
        f = _t(feats).requires_grad_(True)
        x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), f])
        il = O.input_layer(coords, feats, 4)
        kind = trial % 3
        if kind == 0:
            mod = scn.SubmanifoldConvolution(3, nIn, nOut, 3, bool(trial & 1)).to(DEV)
            rb = O.submanifold_rules(il["coords"], [3, 3, 3])
            n_out_rows, in_col = il["V"], 0
        else:
            mod = scn.Convolution(3, nIn, nOut, [2, 2, 2], [2, 2, 2], False).to(DEV)
            rb, oc = O.convolution_rules(il["coords"], [2, 2, 2], [2, 2, 2], np.array([8, 8, 4]))
            n_out_rows, in_col = oc.shape[0], 0
        if hasattr(mod, "bias"):
            mod.bias.data.normal_()
        y = mod(x)
        W = mod.weight.detach().cpu().numpy().reshape(rb.vol, nIn, nOut)
        bias = mod.bias.detach().cpu().numpy() if hasattr(mod, "bias") else None
        ref, _ = O.conv_fwd(il["out"], W, rb, n_out_rows, bias, in_col=in_col)
        tag = "trial %d: %s %d->%d, %d points" % (trial, type(mod).__name__, nIn, nOut, n)
        np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4,
                                   atol=3e-6 * max(np.abs(ref).max(), 1e-3) * nIn, err_msg=tag)
        g = rng.standard_normal(ref.shape).astype(np.float32)
        y.features.backward(_t(g))
        d_in, dW, db = O.conv_bwd(il["out"], g, W, rb, in_col=in_col, want_bias=bias is not None)
        np.testing.assert_allclose(f.grad.cpu().numpy(), O.input_layer_bwd(il, d_in), rtol=1e-4,
                                   atol=3e-6 * max(np.abs(d_in).max(), 1e-3) * nOut, err_msg=tag)
        np.testing.assert_allclose(mod.weight.grad.cpu().numpy().reshape(W.shape), dW, rtol=2e-4,
                                   atol=2e-5 * max(np.abs(dW).max(), 1e-3), err_msg=tag)
        if bias is not None:
            np.testing.assert_allclose(mod.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4, err_msg=tag)


def test_fused_sparse_roi_align_equals_the_dense_two_step_form():
    """ROIAlignRotated3D gathers from the sparse rows through a cell map by default; with `fused = False` it
    densifies, crops and calls `_C.roi_align_rotated_3d_*` like the reference.  Forward must agree bit for
    bit (same statements, zeros for inactive cells); backward up to fp32 summation order (both use atomics)."""
    scn = _scn()
    from maskrcnn_benchmark.layers.roi_align_rotated_3d import ROIAlignRotated3D
    rng = np.random.default_rng(11)
    n = 4000
    coords = np.unique(np.stack([rng.integers(0, 40, n), rng.integers(0, 33, n), rng.integers(0, 7, n),
                                 np.sort(rng.integers(0, 3, n))], 1), axis=0)
    coords = coords[np.argsort(coords[:, 3], kind="stable")].astype(np.int64)
    feats = rng.standard_normal((coords.shape[0], 160)).astype(np.float32)  # 160 planes: two plane groups
    nroi = 70
    rois = np.stack([rng.integers(0, 3, nroi), rng.uniform(-2, 42, nroi), rng.uniform(-2, 35, nroi),
                     rng.uniform(-1, 8, nroi), rng.uniform(0.3, 15, nroi), rng.uniform(0.3, 15, nroi),
                     rng.uniform(0.3, 5, nroi), rng.uniform(-180, 180, nroi)], 1).astype(np.float32)
    for out_size, sampling in (((7, 7, 3), 2), ((5, 6, 11), 0)):  # 330 bins: several LDS passes; adaptive grid
        res = []
        for fused in (True, False):
            f = _t(feats).requires_grad_(True)
            x = scn.InputLayer(3, [48, 48, 8], mode=4)([_t(coords), f])
            layer = ROIAlignRotated3D(out_size, 0.9, sampling)
            layer.fused = fused
            out = layer(x, _t(rois))
            g = torch.Generator(device=DEV).manual_seed(5)
            out.backward(torch.randn(out.shape, device=DEV, generator=g))
            res.append((out.detach(), f.grad.detach()))
        assert res[0][0].shape == (nroi, 160) + out_size
        assert torch.equal(res[0][0], res[1][0])
        torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5)


def test_rpn_proposals_cross_scale_on_device():
    """rpn_glue.rpn_proposals (one top-k + decode + NMS per example over ALL maps, the reference's shape) against
    the numpy / C-oracle restatement; three maps with different strides and anchor sizes, one of them without
    sites for example 1."""
    scn = _scn()
    import sys
    sys.path.insert(0, O.ORACLE_DIR)
    import box_oracle
    import rpn_glue
    rng = np.random.default_rng(23)
    A = 4
    maps, sites, objs, regs, bases, strides = [], [], [], [], [], []
    for mi, (sp, n, st) in enumerate((((64, 64, 8), 2500, (4.0, 4.0, 4.0)), ((32, 32, 4), 900, (8.0, 8.0, 8.0)),
                                      ((16, 16, 1), 120, (16.0, 16.0, 16.0)))):
        b = np.sort(rng.integers(0, 2, n)) if mi < 2 else np.zeros(n, np.int64)
        coords = np.stack([rng.integers(0, sp[0], n), rng.integers(0, sp[1], n), rng.integers(0, sp[2], n), b],
                          1).astype(np.int64)
        x = scn.InputLayer(3, list(sp), mode=3)([_t(coords), _t(np.zeros((n, 1), np.float32))])
        sc = x.get_spatial_locations().numpy()
        V = sc.shape[0]
        base = np.zeros((A, 7), np.float32)
        base[:, 3:6] = [0.2 * (mi + 1), 1.5 * (mi + 1), 2.6]
        base[:, 6] = [0.0, -1.57, -0.785, 0.785]
        maps.append(x)
        sites.append(sc)
        objs.append(rng.standard_normal(V * A).astype(np.float32))
        regs.append((rng.standard_normal((V * A, 7)) * 0.3).astype(np.float32))
        bases.append(base)
        strides.append(st)
    weights = (1.0, 1.0, 1.0, 2.0, 2.0, 2.0, 1.5)
    res = rpn_glue.rpn_proposals(maps, [_t(o) for o in objs], [_t(r) for r in regs], [torch.as_tensor(b) for b in bases],
                                 strides, 20.0, 600, 150, 0.5, (0.3, 0.3), weights, 10000.0)
    want = box_oracle.rpn_proposals(sites, objs, regs, bases, strides, 20.0, O.boxes_iou_3d, O.nms_from_matrix, 600,
                                    150, 0.5, (0.3, 0.3), weights, 10000.0)
    assert len(res) == len(want) == 2
    for (bd, sd), (bw, sw, _) in zip(res, want):
        assert bd.shape[0] == bw.shape[0] > 10
        np.testing.assert_allclose(bd.cpu().numpy(), bw, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(sd.cpu().numpy(), sw, rtol=1e-5, atol=1e-6)
    # the whole batch in one go (one top-k over the padded matrix, one library call for decode + NMS of every
    # example): the same proposals as example by example
    import _hip
    before = []
    lib = _hip.load()
    resb = rpn_glue.rpn_proposals(maps, [_t(o) for o in objs], [_t(r) for r in regs],
                                  [torch.as_tensor(b) for b in bases], strides, 20.0, 600, 150, 0.5, (0.3, 0.3), weights,
                                  10000.0, batch_size=2, batched=True)
    assert len(resb) == 2
    for (bd, sd), (be, se) in zip(res, resb):
        assert torch.equal(bd, be) and torch.equal(sd, se)
    # an example with fewer anchors than pre_nms_top_n: the batched form steps aside (same result as before)
    resc = rpn_glue.rpn_proposals(maps, [_t(o) for o in objs], [_t(r) for r in regs],
                                  [torch.as_tensor(b) for b in bases], strides, 20.0, 100000, 150, 0.5, (0.3, 0.3),
                                  weights, 10000.0, batch_size=2, batched=True)
    resd = rpn_glue.rpn_proposals(maps, [_t(o) for o in objs], [_t(r) for r in regs],
                                  [torch.as_tensor(b) for b in bases], strides, 20.0, 100000, 150, 0.5, (0.3, 0.3),
                                  weights, 10000.0)
    for (bd, sd), (be, se) in zip(resc, resd):
        assert torch.equal(bd, be) and torch.equal(sd, se)


def test_input_layer_prepare_is_matched_by_identity_and_works_for_int32_coords():
    """prepared geometry used to be matched by (data_ptr, n) of a temporary; int32 coordinates tripped
    the 'already holds an input layer' assertion and a recycled address could match a stale entry."""
    scn = _scn()
    rng = np.random.default_rng(4)
    coords, feats = _rand_scene(rng, 5000, (50, 40, 12), 2, 5)
    side = torch.cuda.Stream()
    for dt in (torch.int64, torch.int32):
        c = _t(coords).to(dt)
        layer = scn.InputLayer(3, [50, 40, 12], mode=4)
        layer.prepare(c, torch.device(DEV), side)
        other = _t(coords[:100]).to(dt)                 # an unrelated tensor never consumes the prepared entry
        y_other = layer([other, _t(feats[:100])])
        assert y_other.features.shape[0] <= 100 and len(layer._prepared) == 1
        y = layer([c, _t(feats)])
        assert len(layer._prepared) == 0
        ref = scn.InputLayer(3, [50, 40, 12], mode=4)([_t(coords), _t(feats)])
        assert torch.equal(y.features, ref.features)
        assert torch.equal(y.get_spatial_locations(), ref.get_spatial_locations())


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_input_layer_long_chains_and_chunk_boundaries(mode):
    """voxel scatter (csrc/voxel_scatter.hip): voxels with far more points than the in-register chain sort holds
    (300 points in one voxel), first points spread over many look-back chunks, duplicates that straddle chunk
    boundaries -- sites, rule table and features bit-exact against the oracle"""
    scn = _scn()
    rng = np.random.default_rng(40 + mode)
    n = 9000                                             # 9 chunks of 1024
    coords = np.stack([rng.integers(0, 30, n), rng.integers(0, 30, n), rng.integers(0, 4, n),
                       np.sort(rng.integers(0, 2, n))], 1).astype(np.int64)
    hot = rng.choice(n, 300, replace=False)
    coords[hot, :3] = (7, 7, 1)                          # one voxel (per batch index) with ~150 points each
    feats = rng.standard_normal((n, 5)).astype(np.float32)
    layer = scn.InputLayer(3, [32, 32, 4], mode=mode)
    f = _t(feats).requires_grad_(True)
    x = layer([_t(coords), f])
    ref = O.input_layer(coords, feats, mode)
    assert x.metadata.input["V"] == ref["V"] and ref["max_active"] > 100 or mode in (1, 2)
    np.testing.assert_array_equal(x.get_spatial_locations().numpy(), ref["coords"])
    np.testing.assert_array_equal(x.metadata.input["point_site"].cpu().numpy(), ref["point_voxel"])
    hdr, rules = x.metadata.inputLayerRuleBook()
    np.testing.assert_array_equal(rules.cpu().numpy(), ref["rules"])
    np.testing.assert_array_equal(x.features.detach().cpu().numpy(), ref["out"])
    g = rng.standard_normal(ref["out"].shape).astype(np.float32)
    x.features.backward(_t(g))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), O.input_layer_bwd(ref, g))


def test_output_layer_and_incremental_input_match_oracle():
    """OutputLayer (pybind.cpp:163-170) = InputLayer backwards without averaging; Metadata.setInputSpatialLocations
    builds the same grid as an InputLayer over the unique locations"""
    scn = _scn()
    rng = np.random.default_rng(77)
    coords, feats = _rand_scene(rng, 4000, (12, 10, 6), 2, 6)
    for mode in (1, 2, 3, 4):
        f = _t(feats).requires_grad_(True)
        x = scn.InputLayer(3, [16, 16, 8], mode=mode)([_t(coords), f])
        out = scn.OutputLayer(3)(x)
        il = O.input_layer(coords, feats, mode)
        il_noavg = dict(il, mode=3 if mode == 4 else mode)
        want = O.input_layer_bwd(il_noavg, il["out"])            # InputLayer_BackwardPass(average=false)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        g = rng.standard_normal(want.shape).astype(np.float32)
        out.backward(_t(g))
        # d features = InputLayer backward of (OutputLayer backward of g) = bwd(fwd_noavg(g))
        d_sites = np.zeros_like(il["out"])
        O.lib().oracle_input_layer_fwd(g, d_sites, il["V"], il["max_active"], g.shape[1], il["rules"], 0)
        np.testing.assert_array_equal(f.grad.cpu().numpy(), O.input_layer_bwd(il, d_sites))
    # incremental construction
    md = scn.Metadata(3)
    md.setInputSpatialSize(torch.LongTensor([16, 16, 8]))
    hf = torch.FloatTensor()
    md.setInputSpatialLocations(hf, torch.as_tensor(coords), torch.as_tensor(feats), False)   # keep first
    il1 = O.input_layer(coords, feats, 1)
    np.testing.assert_array_equal(hf.numpy(), il1["out"])
    xin = scn.SparseConvNetTensor(hf.to(DEV), md, torch.LongTensor([16, 16, 8]))
    conv = scn.SubmanifoldConvolution(3, 6, 8, 3, False).to(DEV)
    y = conv(xin)
    ref = scn.InputLayer(3, [16, 16, 8], mode=1)([_t(coords), _t(feats)])
    torch.testing.assert_close(y.features, conv(ref).features, rtol=0, atol=0)
    np.testing.assert_array_equal(xin.get_spatial_locations().numpy(), il1["coords"])


def test_prefetched_geometry_is_parked_with_one_event_and_reaped():
    """SCN.Metadata_3.hand_over: geometry built on another stream and consumed on the current one is kept until its
    Metadata dies, then parked with ONE event on the consumer stream and dropped once that event has passed
    (replaces tensor.record_stream per tensor: ~200 marker packets per pass)."""
    import gc
    import sparseconvnet as scn
    from sparseconvnet import SCN
    rng = np.random.default_rng(3)
    coords = np.stack([rng.integers(0, 30, 3000), rng.integers(0, 30, 3000), rng.integers(0, 8, 3000),
                       np.zeros(3000, np.int64)], 1).astype(np.int64)
    feats = rng.standard_normal((3000, 4)).astype(np.float32)
    layer = scn.InputLayer(3, [32, 32, 8], mode=4)
    conv = scn.SubmanifoldConvolution(3, 4, 8, 3, False).to(DEV)
    c = _t(coords)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    SCN._reap_handed_over()
    n0 = len(SCN._parked)
    layer.prepare(c, torch.device(DEV), side)
    x = layer([c, _t(feats)])
    md = x.metadata
    assert getattr(md, "_handed_over", None) is not None and md._handed_over[0] == torch.cuda.current_stream()
    with torch.no_grad():                              # (an autograd graph would keep the Metadata alive through its nodes)
        y = conv(x).features.clone()
    assert len(SCN._parked) == n0                      # alive: nothing parked yet
    del x, md
    gc.collect()
    assert len(SCN._parked) == n0 + 1                  # parked with one event
    torch.cuda.synchronize()
    SCN._reap_handed_over()
    assert len(SCN._parked) == n0                      # the consumer stream has passed it
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("groups,nIn,nOut", [(2, 32, 64), (4, 64, 32), (3, 12, 6)])
def test_grouped_convolutions_match_block_diagonal_oracle(groups, nIn, nOut):
    """`groups` > 1 (VERDICT r3 missing #4; the reference carries it through every kernel: weight [vol, groups, nIn/g,
    nOut/g], planes group-major, SCN/CPU/Convolution.cpp:8-43,139-147): SubmanifoldConvolution, Convolution and
    Deconvolution forward + backward (with bias) equal the ungrouped operator with the block-diagonal weight."""
    import oracle_lib as O
    scn = _scn()
    rng = np.random.default_rng(groups * 100 + nIn)
    coords = np.stack([rng.integers(0, s, 1500) for s in (12, 12, 6)] + [np.sort(rng.integers(0, 2, 1500))], 1).astype(np.int64)
    feats = rng.standard_normal((1500, nIn)).astype(np.float32)
    il = O.input_layer(coords, feats, 4)
    ip, op = nIn // groups, nOut // groups

    def dense(w):      # [vol, g, ip, op] -> block-diagonal [vol, nIn, nOut]
        W = np.zeros((w.shape[0], nIn, nOut), np.float32)
        for g in range(groups):
            W[:, g * ip:(g + 1) * ip, g * op:(g + 1) * op] = w[:, g]
        return W

    x = scn.InputLayer(3, [16, 16, 8], mode=4)([_t(coords), _t(feats).requires_grad_(True)])
    leaf = x.features.detach().clone().requires_grad_(True)
    xs = scn.SparseConvNetTensor()
    xs.metadata, xs.spatial_size, xs.features = x.metadata, x.spatial_size, leaf
    # submanifold, with bias
    conv = scn.SubmanifoldConvolution(3, nIn, nOut, 3, True, groups).to(DEV)
    conv.bias.data.normal_()
    assert tuple(conv.weight.shape) == (27, groups, ip, op)
    y = conv(xs)
    rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    W = dense(conv.weight.detach().cpu().numpy())
    ref, macs = O.conv_fwd(il["out"], W, rb, il["V"], conv.bias.detach().cpu().numpy())
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * nIn)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g))
    d_in, dW, db = O.conv_bwd(il["out"], g, W, rb, want_bias=True)
    np.testing.assert_allclose(leaf.grad.cpu().numpy(), d_in, rtol=1e-4, atol=1e-5 * nOut)
    gw = conv.weight.grad.cpu().numpy()
    for k in range(groups):
        np.testing.assert_allclose(gw[:, k], dW[:, k * ip:(k + 1) * ip, k * op:(k + 1) * op], rtol=1e-4,
                                   atol=1e-5 * np.abs(dW).max())
    np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4)
    # strided convolution then its transpose
    down = scn.Convolution(3, nIn, nOut, 2, 2, False, groups).to(DEV)
    up = scn.Deconvolution(3, nOut, nIn, 2, 2, False, groups).to(DEV)
    leaf2 = x.features.detach().clone().requires_grad_(True)
    xs.features = leaf2
    z = up(down(xs))
    rbs, oc = O.convolution_rules(il["coords"], [2, 2, 2], [2, 2, 2], [8, 8, 4])
    Wd = dense(down.weight.detach().cpu().numpy())
    Wu = np.zeros((8, nOut, nIn), np.float32)
    wu = up.weight.detach().cpu().numpy()
    for k in range(groups):
        Wu[:, k * op:(k + 1) * op, k * ip:(k + 1) * ip] = wu[:, k]
    mid, _ = O.conv_fwd(il["out"], Wd, rbs, oc.shape[0])
    back, _ = O.conv_fwd(mid, Wu, rbs, il["V"], in_col=1)
    np.testing.assert_allclose(z.features.detach().cpu().numpy(), back, rtol=1e-4, atol=1e-4 * max(nIn, nOut))
    g2 = rng.standard_normal(back.shape).astype(np.float32)
    z.features.backward(_t(g2))
    d_mid, dWu, _ = O.conv_bwd(mid, g2, Wu, rbs, in_col=1)
    d_x, dWd, _ = O.conv_bwd(il["out"], d_mid, Wd, rbs)
    np.testing.assert_allclose(leaf2.grad.cpu().numpy(), d_x, rtol=1e-4, atol=1e-4 * max(nIn, nOut))
    gd = down.weight.grad.cpu().numpy()
    for k in range(groups):
        np.testing.assert_allclose(gd[:, k], dWd[:, k * ip:(k + 1) * ip, k * op:(k + 1) * op], rtol=1e-4,
                                   atol=1e-5 * np.abs(dWd).max())


def test_fused_cross_scale_topk_matches_torch_topk():
    """aabr_rpn_topk_maps (csrc/iou_nms.hip): the per-example `objectness.topk(pre_nms_top_n, sorted=True)` of
    RPNPostProcessor (rpn/inference_3d.py:107-112) for all examples in four launches, over the maps' logit vectors through
    the segment table (nothing concatenated) -- against torch.topk on the concatenated copy: same values in the same order,
    same index set, equal logits by ascending index; k > half the list, k = the whole list, an empty example, negative /
    zero / repeated logits; > 4096 exact ties at the cut are reported, and rpn_proposals then falls back."""
    import _hip
    from _hip import ptr, stream, check
    lib = _hip.load()
    rng = np.random.default_rng(77)
    A, n_maps, nb = 4, 3, 4
    counts = [[5000, 3000, 0, 700], [800, 0, 0, 90], [64, 10, 0, 3]]       # sites per (map, example)
    obj = []
    for m in range(n_maps):
        v = rng.standard_normal(sum(counts[m]) * A).astype(np.float32)
        v[rng.integers(0, v.size, v.size // 50)] = 0.0                      # repeated values (exact ties)
        v[rng.integers(0, v.size, v.size // 80)] = 1.25
        obj.append(_t(v))
    segs, sites, s0 = [], [], [0] * n_maps
    for bi in range(nb):
        seg = [0]
        for m in range(n_maps):
            seg.append(seg[-1] + counts[m][bi] * A)
        segs += seg
        sites += s0
        s0 = [s0[m] + counts[m][bi] for m in range(n_maps)]
    ns = [segs[bi * (n_maps + 1) + n_maps] for bi in range(nb)]
    for kreq in (2000, 600, 1):
        ks = [min(kreq, n) for n in ns]
        kmax = max(ks)
        sel = torch.full((nb, kmax), -1, dtype=torch.int64, device=DEV)
        info = torch.full((nb, 2), -1, dtype=torch.int32, device=DEV)
        scr = torch.empty(int(lib.aabr_rpn_topk_scratch_words(nb)) + 2, dtype=torch.int32, device=DEV)
        off = (-scr.data_ptr() // 4) % 2
        check(lib.aabr_rpn_topk_maps(n_maps, _hip.ptrs(obj), nb, _hip.i32xn(segs), _hip.i32xn(sites), A, _hip.i32xn(ks),
                                     ptr(sel), kmax, ptr(info), scr.data_ptr() + 4 * off, stream()))
        inf = info.cpu().numpy()
        st = [0] * n_maps
        for bi in range(nb):
            cat = torch.cat([obj[m][st[m] * A:(st[m] + counts[m][bi]) * A] for m in range(n_maps)])
            st = [st[m] + counts[m][bi] for m in range(n_maps)]
            k = ks[bi]
            assert inf[bi, 1] == 0 and inf[bi, 0] >= k
            if k == 0:
                continue
            got = sel[bi, :k]
            tv, ti = cat.topk(k, sorted=True)
            assert torch.equal(cat[got], tv)                                  # the same values, in the same order
            # equal logits come out by ascending index; the selection at the cut takes the lowest indices
            c = cat.cpu().numpy()
            order = np.lexsort((np.arange(c.size), -c.astype(np.float64)))
            np.testing.assert_array_equal(got.cpu().numpy(), order[:k])
    # > 4096 exactly equal logits at the cut: reported
    flat = [_t(np.full(sum(counts[m]) * A, 0.5, np.float32)) for m in range(n_maps)]
    ks = [min(2000, n) for n in ns]
    sel = torch.full((nb, 3000), -1, dtype=torch.int64, device=DEV)        # (the loop's last `sel` holds one column)
    check(lib.aabr_rpn_topk_maps(n_maps, _hip.ptrs(flat), nb, _hip.i32xn(segs), _hip.i32xn(sites), A, _hip.i32xn(ks), ptr(sel),
                                 max(ks), ptr(info), scr.data_ptr() + 4 * off, stream()))
    inf = info.cpu().numpy()
    assert inf[0, 1] == 1 and inf[1, 1] == 1 and inf[2, 1] == 0 and inf[3, 1] == 0     # 35 k / 12 k tied; empty; 3,172 tied: sorted
    assert lib.aabr_rpn_topk_maps(n_maps, _hip.ptrs(flat), nb, _hip.i32xn(segs), _hip.i32xn(sites), A, _hip.i32xn([3000] * nb),
                                  ptr(sel), 3000, ptr(info), scr.data_ptr() + 4 * off, stream()) != 0   # k <= 2048


def test_rpn_proposals_fall_back_when_the_cut_is_tied_en_masse():
    """all logits equal (an untrained head can do that): the fused top-k reports the overflow and rpn_proposals selects that
    example with torch.topk instead -- same proposals as the round-4 path"""
    scn = _scn()
    import rpn_glue
    rng = np.random.default_rng(5)
    n, sp, A = 3000, (64, 64, 8), 4
    coords = np.stack([rng.integers(0, sp[0], n), rng.integers(0, sp[1], n), rng.integers(0, sp[2], n),
                       np.zeros(n, np.int64)], 1).astype(np.int64)
    x = scn.InputLayer(3, list(sp), mode=3)([_t(coords), _t(np.zeros((n, 1), np.float32))])
    V = x.features.shape[0]
    base = np.zeros((A, 7), np.float32)
    base[:, 3:6] = [0.2, 1.5, 2.6]
    base[:, 6] = [0.0, -1.57, -0.785, 0.785]
    obj = _t(np.full(V * A, 0.25, np.float32))
    reg = _t((rng.standard_normal((V * A, 7)) * 0.3).astype(np.float32))
    args = ([x], [obj], [reg], [torch.as_tensor(base)], [(4.0, 4.0, 4.0)], 20.0, 600, 150, 0.5, (0.3, 0.3))
    f0 = rpn_glue.topk_stats["fallbacks"]
    a = rpn_glue.rpn_proposals(*args)
    assert rpn_glue.topk_stats["fallbacks"] == f0 + 1
    rpn_glue.fused_topk = False
    try:
        b = rpn_glue.rpn_proposals(*args)
    finally:
        rpn_glue.fused_topk = True
    assert len(a) == len(b) == 1 and a[0][0].shape == b[0][0].shape and a[0][0].shape[0] > 0

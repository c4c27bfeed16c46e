"""The HEADLINE workload against the oracle at its own size (VERDICT r3 "next round" item 1).

bench.py's network path exactly -- `bench.Workload` (seeds 9000+, 4 x S80k @ 2 cm = 319,992 points -> 309,103 sites),
`compiled_graph=True` (one launch list per pass, BatchNorm statistics fused into the wide convolution's write-out,
residual adds fused into their producers, weight gradients on the library's second stream) -- compared with
`ref_net.FpnOracle`, the composition of the oracle kernels that are pinned to the reference's own compiled CPU kernels
(tests/test_oracle_ref_kernels.py; reference: SparseConvNet/sparseconvnet/fpn_net.py:140-203,
SCN/CPU/Convolution.cpp:117-185, SCN/CPU/BatchNormalization.cpp:12-107):

  * fp32 (the reference's arithmetic): six RPN maps, the ROI maps, EVERY BatchNorm output, every running statistic,
    every parameter gradient and the input-feature gradient, with the tolerances of
    test_gpu_fpn.py::test_fpn_net_matches_oracle_composition;
  * bf16 feature storage (extension; BASELINE configs[2] / [4] name bf16): the same against the oracle evaluated in
    the bf16 storage model (FpnOracle(storage="bf16"): every stored activation / activation gradient / packed weight
    rounded to bf16 at the place the device rounds it) with the STATED tolerance of DESIGN.md section 4.

At this size every large layer takes `k_conv_cs` inside the network (>= 320 workgroups), which the 2 x 20k-point test
does not reach."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
import ref_net
import synth_scenes as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- bf16 storage: the stated tolerance ("bf16 storage") ---------------------------------------
# u = 2^-8: half a unit in the last place of an 8-bit significand, the relative error of ONE bf16 store.
# Bounds = 2-4x what was measured on MI355X at this size (round 4: in brackets).
BF16_U = 2.0 ** -8
# (a) teacher-forced -- the oracle's BatchNorm outputs are replaced by the device's stored ones, so every compared
#     tensor is at most a convolution (+ add) + BatchNorm away from bit-identical inputs: a stored value differs by
#     the occasional one-ulp flip of the last 1-2 stores (fp32 vs double accumulation landing on different sides of a
#     rounding boundary), amplified by the BatchNorm scale: relative L2 error of a tensor <= 1 u [0.24 u], largest
#     error <= 4 u of the tensor's largest value [1.26 u].
BF16_TF_L2 = 1 * BF16_U
BF16_TF_MAX = 4 * BF16_U
# (b) free-running -- both sides run all ~100 layers on their own roundings, ReLU masks flip where an activation is
#     within a rounding of 0: relative L2 error of each returned map <= 10 u = 3.9e-2 [3.4-6.4 u].
BF16_FREE_L2 = 10 * BF16_U
# (c) gradients (teacher-forced forward, free-running backward: ~100 stored activation gradients in a chain, each one
#     rounding): relative L2 error of every parameter-gradient tensor and of the input gradient <= 8 u = 3.1e-2
#     [4.2 u / 3.4 u], cosine >= 0.9995 [0.99987].
BF16_GRAD_L2 = 8 * BF16_U
BF16_GRAD_COS = 0.9995


def _relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float((a * b).sum() / (np.sqrt((a * a).sum() * (b * b).sum()) + 1e-300))


def _oracle_net(P, storage="f32"):
    return ref_net.FpnOracle(P, (4096, 4096, 512), [[2, 2, 2]] * 8, [[2, 2, 2]] * 8,
                             [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]], storage=storage)


class _Rows(object):
    """device rows <-> oracle rows, level by level.  With the reference's site order ("first_seen") the two lists are
    identical and every map is compared as it is; with brick-major order the device's rows of a sample are a permutation
    (SURVEY.md 7): rows are matched by their coordinates and everything per-site is compared through that match."""

    def __init__(self, md, oracle_sites, exact):
        self.md, self.sites, self.exact, self.perm = md, oracle_sites, exact, {}

    def of(self, spatial):
        sp = tuple(int(v) for v in spatial)
        p = self.perm.get(sp)
        if p is None:
            dev = self.md.getSpatialLocations(torch.LongTensor(list(sp))).numpy()
            ref = self.sites[sp]
            if self.exact:
                np.testing.assert_array_equal(dev, ref)                       # site lists: exact
                p = np.arange(len(ref))
            else:
                key = lambda c: ((c[:, 3].astype(np.int64) * 8192 + c[:, 0]) * 8192 + c[:, 1]) * 8192 + c[:, 2]
                kd, kr = key(dev), key(ref)
                assert len(kd) == len(kr)
                o = np.argsort(kr)
                pos = np.searchsorted(kr[o], kd)
                assert (pos < len(kr)).all() and (kr[o][pos] == kd).all(), "site sets differ at %s" % (sp,)
                p = o[pos]
                assert len(np.unique(p)) == len(p)
            self.perm[sp] = p
        return p

    def to_dev(self, spatial, a_oracle):
        """oracle-ordered rows -> the device's order"""
        return a_oracle[self.of(spatial)]

    def to_oracle(self, spatial, a_dev):
        out = np.empty_like(a_dev)
        out[self.of(spatial)] = a_dev
        return out


def _bench_pass(dtype, order="first_seen", config=2):
    """one forward + backward of bench.Workload's network path on its own first batch; returns everything compared"""
    sys.path.insert(0, REPO)
    import bench
    import dp
    import sparseconvnet as scn
    from sparseconvnet import planExecutor
    assert planExecutor.dw_side_stream and planExecutor.fuse_adds and planExecutor.conv_bn_stats   # the bench's switches
    saved = (bench.SCENES_PER_STEP, bench.N_POINTS, os.environ.get("AABR_BENCH_SITE_ORDER"))
    if config == 4:                      # what `bench.py --config 4` sets: one 1.5 M-point scene per step
        bench.SCENES_PER_STEP, bench.N_POINTS = 1, 1500000
    os.environ["AABR_BENCH_SITE_ORDER"] = order
    try:
        wl = bench.Workload(scn, torch, dp, torch.device(DEV), dtype, 0, 1, 1)
        n_pts = bench.N_POINTS
    finally:
        bench.SCENES_PER_STEP, bench.N_POINTS = saved[0], saved[1]
        if saved[2] is None:
            del os.environ["AABR_BENCH_SITE_ORDER"]
        else:
            os.environ["AABR_BENCH_SITE_ORDER"] = saved[2]
    net = wl.net
    assert net.compiled_graph and net.voxel_scale == bench.VOXEL_SCALE and net.site_order == order
    locs_t, feats_t = wl.batches[0]
    locs, feats = locs_t.cpu().numpy(), feats_t.detach().cpu().numpy()
    # the same scenes bench.py times: seeds 9000.., 2 cm
    l0, f0 = S.make_scene(n_pts, 9000, bench.VOXEL_SCALE)
    assert (locs[:l0.shape[0], :3] == l0).all() and (feats[:l0.shape[0]] == f0).all()
    P = ref_net.fpn_params(net)
    bn_mods = ref_net.fpn_bn_modules(net)
    wl.flat.zero_grad()
    planExecutor.debug_passes = []
    before = planExecutor.stats["passes"]
    from sparseconvnet import SCN
    tr, SCN.trace = SCN.trace, None
    try:
        rpn_maps, roi_maps = net([locs_t, feats_t])
        assert planExecutor.stats["passes"] == before + 1          # the compiled graph ran, not the modules
        ps = planExecutor.debug_passes[-1]
    finally:
        planExecutor.debug_passes = None
        SCN.trace = tr
    md = rpn_maps[0].metadata
    assert md.site_order == order
    by_mod = ps.bn_outputs()
    acts = {name: by_mod[m].detach().float().cpu().numpy() for name, m in bn_mods.items()}
    rng = np.random.default_rng(3)
    # output gradients are drawn in the ORACLE's row order and handed to the device through the row match (below)
    return dict(net=net, wl=wl, locs=locs, feats=feats, P=P, bn_mods=bn_mods, acts=acts, rpn=rpn_maps, roi=roi_maps,
                rng=rng, feats_t=feats_t, md=md, order=order, ps=ps)


def _backward(r, rows):
    """draw the output gradients (oracle row order), run the device's backward with them; returns the oracle-order list"""
    G = [r["rng"].standard_normal(m.features.shape).astype(np.float32) / m.features.shape[0] for m in r["rpn"]]
    torch.autograd.backward([m.features for m in r["rpn"]],
                            [torch.as_tensor(rows.to_dev(m.spatial_size.tolist(), g)).to(DEV) for m, g in zip(r["rpn"], G)])
    torch.cuda.synchronize()
    r["d_feats"] = r["feats_t"].grad.detach().cpu().numpy()
    return G


def _check_fp32(r, min_sites):
    net, P, acts = r["net"], r["P"], r["acts"]
    O.set_threads(16)
    fo = _oracle_net(P)
    o_rpn, o_roi = fo.forward(r["locs"], r["feats"])
    assert fo.il["V"] == r["md"].input["V"] > min_sites
    rows = _Rows(r["md"], fo.sites, r["order"] == "first_seen")
    for i, (d, o) in enumerate(zip(r["rpn"], o_rpn)):
        assert tuple(d.spatial_size.tolist()) == o.spatial
        assert _relerr(d.features.detach().cpu().numpy(), rows.to_dev(o.spatial, o.v)) < 2e-3, i
    for d, o in zip(r["roi"], o_roi):
        assert _relerr(d.features.detach().cpu().numpy(), rows.to_dev(o.spatial, o.v)) < 2e-3
    # every BatchNorm output of the pass (read from the compiled graph's arena), ReLU masks flip only at rounding
    # distance of 0
    assert len(acts) == 34
    flips, worst = 0, 0.0
    for name, a in acts.items():
        want = rows.to_dev(fo.act_spatial[name], fo.acts[name])
        e = _relerr(a, want)
        worst = max(worst, e)
        assert e < 2e-3, name
        flips += int(((a > 0) != (want > 0)).sum())
    assert flips <= 1e-4 * sum(a.size for a in acts.values()) + 4
    # every running statistic (momentum 0.95, unbiased variance)
    for name, m in r["bn_mods"].items():
        np.testing.assert_allclose(m.running_mean.cpu().numpy(), P[name]["running_mean_out"], rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(m.running_var.cpu().numpy(), P[name]["running_var_out"], rtol=2e-3, atol=2e-5)
    # ---- backward: the oracle replays its forward on the device's BatchNorm outputs (identical ReLU masks)
    G = _backward(r, rows)
    P2 = {k: (dict(v) if isinstance(v, dict) else v) for k, v in P.items()}
    fo2 = _oracle_net(P2)
    fo2.override = {name: rows.to_oracle(fo.act_spatial[name], a) for name, a in acts.items()}
    fo2.forward(r["locs"], r["feats"])
    grads = fo2.backward(G)
    names = ref_net.fpn_param_names(net)
    checked, gworst = 0, 0.0
    for key, par in names.items():
        if key not in grads:
            assert par.grad is None or float(par.grad.abs().max()) == 0.0, key     # dead branches (ups 5..8)
            continue
        got = par.grad.detach().cpu().numpy().reshape(grads[key].shape)
        e = _relerr(got, grads[key])
        gworst = max(gworst, e)
        assert e < 3e-3, key
        checked += 1
    assert checked >= 100
    e_in = _relerr(r["d_feats"], grads["d_feats"])
    assert e_in < 3e-3
    print("full-size fp32 (%s rows, %d sites) vs oracle: worst BN output %.2e, mask flips %d, worst parameter gradient "
          "%.2e over %d tensors, input gradient %.2e" % (r["order"], fo.il["V"], worst, flips, gworst, checked, e_in))


@pytest.mark.parametrize("order", ["first_seen", "brick"])
def test_bench_path_fp32_matches_oracle_at_full_size(order):
    _check_fp32(_bench_pass(torch.float32, order), 300000)


def _check_bf16(r, min_sites):
    net, P, acts = r["net"], r["P"], r["acts"]
    O.set_threads(16)
    fails = []

    def need(ok, what):
        if not ok:
            fails.append(what)

    # (b) free-running oracle in the bf16 storage model
    fo = _oracle_net(P, "bf16")
    o_rpn, o_roi = fo.forward(r["locs"], r["feats"])
    assert fo.il["V"] == r["md"].input["V"] > min_sites
    rows = _Rows(r["md"], fo.sites, r["order"] == "first_seen")
    free = []
    for i, (d, o) in enumerate(zip(r["rpn"] + r["roi"], o_rpn + o_roi)):
        assert d.features.dtype == torch.float32
        free.append(_l2(d.features.detach().cpu().numpy(), rows.to_dev(o.spatial, o.v)))
        need(free[-1] <= BF16_FREE_L2, ("free map", i, free[-1]))
    # (a) teacher-forced: the oracle's BatchNorm outputs replaced by the device's
    P2 = {k: (dict(v) if isinstance(v, dict) else v) for k, v in P.items()}
    fo2 = _oracle_net(P2, "bf16")
    fo2.override = {name: rows.to_oracle(fo.act_spatial[name], a) for name, a in acts.items()}
    t_rpn, t_roi = fo2.forward(r["locs"], r["feats"])
    tf_l2, tf_max = 0.0, 0.0
    for name, a in acts.items():
        ref = rows.to_dev(fo2.act_spatial[name], fo2.acts[name])   # from teacher-forced inputs, before the override
        l2, mx = _l2(a, ref), _relerr(a, ref)
        tf_l2, tf_max = max(tf_l2, l2), max(tf_max, mx)
        need(l2 <= BF16_TF_L2 and mx <= BF16_TF_MAX, ("teacher-forced", name, l2, mx))
    for i, (d, o) in enumerate(zip(r["rpn"] + r["roi"], t_rpn + t_roi)):
        want = rows.to_dev(o.spatial, o.v)
        l2, mx = _l2(d.features.detach().cpu().numpy(), want), _relerr(d.features.detach().cpu().numpy(), want)
        tf_l2, tf_max = max(tf_l2, l2), max(tf_max, mx)
        need(l2 <= BF16_TF_L2 and mx <= BF16_TF_MAX, ("teacher-forced map", i, l2, mx))
    rs = 0.0
    for name, m in r["bn_mods"].items():
        for got, want in ((m.running_mean.cpu().numpy(), P2[name]["running_mean_out"]),
                          (m.running_var.cpu().numpy(), P2[name]["running_var_out"])):
            e = float(np.max(np.abs(got - want) / (5e-3 * np.abs(want) + 1e-4)))
            rs = max(rs, e)
            need(e <= 1.0, ("running statistic", name, e))
    # (c) gradients
    G = _backward(r, rows)
    grads = fo2.backward(G)
    names = ref_net.fpn_param_names(net)
    checked, g_l2, g_cos = 0, 0.0, 1.0
    for key, par in names.items():
        if key not in grads:
            assert par.grad is None or float(par.grad.abs().max()) == 0.0, key
            continue
        assert par.grad.dtype == torch.float32
        got = par.grad.detach().cpu().numpy().reshape(grads[key].shape)
        l2, c = _l2(got, grads[key]), _cos(got, grads[key])
        g_l2, g_cos = max(g_l2, l2), min(g_cos, c)
        need(l2 <= BF16_GRAD_L2 and c >= BF16_GRAD_COS, ("gradient", key, l2, c))
        checked += 1
    assert checked >= 100
    l2_in, c_in = _l2(r["d_feats"], grads["d_feats"]), _cos(r["d_feats"], grads["d_feats"])
    need(l2_in <= BF16_GRAD_L2 and c_in >= BF16_GRAD_COS, ("input gradient", l2_in, c_in))
    print("full-size bf16 (%s rows, %d sites) vs bf16-model oracle: free-running map L2 %s (bound %.2e); teacher-forced "
          "worst L2 %.2e (bound %.2e), worst max-error %.2e (bound %.2e); running statistics worst %.2f of their tolerance; "
          "parameter gradients worst L2 %.2e (bound %.2e), worst cosine %.5f over %d tensors; input gradient L2 %.2e cosine "
          "%.5f" % (r["order"], fo.il["V"], ["%.2e" % v for v in free], BF16_FREE_L2, tf_l2, BF16_TF_L2, tf_max,
                     BF16_TF_MAX, rs, g_l2, BF16_GRAD_L2, g_cos, checked, l2_in, c_in))
    assert not fails, fails[:12]


@pytest.mark.parametrize("order", ["first_seen", "brick"])
def test_bench_path_bf16_matches_bf16_oracle_at_full_size(order):
    _check_bf16(_bench_pass(torch.bfloat16, order), 300000)


# ---- BASELINE configs[4]: one 1.5 M-point scene @ 2 cm through the same network and step (`bench.py --config 4`) -------
# at this size the 32 -> 32 layers of the finest level dispatch to k_conv_narrow (bf16 storage, >= 400 k
# output rows) -- the kernel and the compiled-graph path of this config had only met the oracle at <= 3,001 points.
def test_config4_bf16_matches_bf16_oracle_at_full_size():
    from sparseconvnet import SCN
    r = _bench_pass(torch.bfloat16, "brick", 4)
    V0 = r["md"].input["V"]
    assert SCN.narrow_ok(32, 32, V0, V0, 27, True), "the 32 -> 32 layers of this config must run k_conv_narrow"
    _check_bf16(r, 800000)


def test_config4_fp32_matches_oracle_at_full_size():
    _check_fp32(_bench_pass(torch.float32, "first_seen", 4), 800000)


def test_bench_proposal_sets_nms_decision_and_survivors():
    """VERDICT r3 weak #3 on the bench's OWN proposal sets: one step of bench.Workload (fp32), the 4 x 2,000 decoded,
    clamped boxes that go into the rotated NMS captured on the way.  (1) count the ordered pairs with pre-filter > 0 on
    different sides of thresh under the A15 IoU and the exact polygon IoU (oracle/clip_oracle.c); (2) the device's
    survivors (decision on clip_iou_exact, csrc/iou_math.h) equal the oracle's greedy loop with the two matrices, and
    the count of proposals the old rule (decision on the A15 value) would have changed is reported."""
    sys.path.insert(0, REPO)
    import bench
    import dp
    import rpn_glue
    import sparseconvnet as scn
    import _nms
    wl = bench.Workload(scn, torch, dp, torch.device(DEV), torch.float32, 0, 1, 1)
    wl.prefetch_geometry = False
    rpn_glue.debug_nms_inputs = []
    try:
        wl.forward_backward(0)
        torch.cuda.synchronize()
        sets = [(b.cpu().numpy(), s.cpu().numpy()) for b, s in rpn_glue.debug_nms_inputs]
    finally:
        rpn_glue.debug_nms_inputs = None
    assert len(sets) == bench.SCENES_PER_STEP
    O.set_threads(16)
    total_pre = total_dis = changed = 0
    for (b7, sc), (boxes, scores) in zip(sets, wl.last[1]):
        assert b7.shape == (2000, 7) and (np.diff(sc) <= 0).all()          # sorted by descending score
        a15 = O.boxes_iou_3d(b7, b7, (0, 0, 0, 0), -1, True)
        ex = O.clip_iou_matrix(b7)
        pre = a15 > 0
        np.fill_diagonal(pre, False)
        total_pre += int(pre.sum())
        total_dis += int((pre & ((a15 >= 0.5) != (ex >= 0.5))).sum())
        order = np.arange(2000, dtype=np.int32)
        want = O.nms_prefilter_decide(a15, ex, order, 0.5)[:1000]
        old = O.nms_from_matrix(a15, order, 0.5)[:1000]
        changed += len(set(want.tolist()) ^ set(old.tolist()))
        got = _nms.rotate_nms_sorted(torch.as_tensor(b7).to(DEV), 0.5, 1000, True).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(scores.cpu().numpy(), sc[want])     # what the step itself kept
    print("bench proposal sets: %d ordered pairs with pre-filter > 0, %d on different sides of 0.5 under A15 vs exact "
          "polygon IoU; survivors that differ between the two rules: %d" % (total_pre, total_dis, changed))

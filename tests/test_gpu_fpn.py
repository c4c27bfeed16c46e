"""Whole-network parity and the full-size configurations of BASELINE.json on a real MI355X.

  * FPN_Net forward + backward against the ORACLE COMPOSITION of the same network (tests/ref_net.py::FpnOracle;
    its kernels are pinned to the reference's own CPU kernels, tests/test_oracle_ref_kernels.py): all six RPN
    maps and every parameter gradient.
  * configs[2]: 4 x S80k @ 2 cm, bf16 feature storage, + cross-scale rotated NMS  -- size-independent properties.
  * configs[4]: one 1.5 M-point scene @ 2 cm through the whole FPN_Net, bf16   -- size-independent properties.
  * bench.py --gpus 2 rehearsal on one GPU (gloo, shared device)."""
import json
import os
import subprocess
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


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _fpn(voxel_scale=20, **kw):
    from test_cabi_and_host import default_fpn
    return default_fpn(**kw)


def _oracle_net(P):
    return ref_net.FpnOracle(P, (4096, 4096, 512), [[2, 2, 2]] * 8, [[2, 2, 2]] * 8,
                             [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]])


def test_fpn_net_matches_oracle_composition():
    """2 x 20k-point scenes @ 5 cm through the default 9-scale backbone: six RPN maps (sites exact, features
    within tolerance) and all parameter gradients against the oracle composition of fpn_net.py:168-203."""
    torch.manual_seed(1)
    net = _fpn().to(DEV)
    locs, feats = S.make_batch(2, 20000, 5, 20)
    P = ref_net.fpn_params(net)          # before the forward (BN running stats as the oracle will see them)
    bn_mods = ref_net.fpn_bn_modules(net)
    acts = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, name=name: acts.__setitem__(
        name, out.features.detach().float().cpu().numpy())) for name, m in bn_mods.items()]
    f = _t(feats).requires_grad_(True)
    rpn_maps, roi_maps = net([_t(locs), f])
    for h in hooks:
        h.remove()
    # ---- forward
    fo = _oracle_net(P)
    o_rpn, o_roi = fo.forward(locs, feats)
    assert len(rpn_maps) == len(o_rpn) == 6
    for i, (d, o) in enumerate(zip(rpn_maps, o_rpn)):
        assert tuple(d.spatial_size.tolist()) == o.spatial
        np.testing.assert_array_equal(d.get_spatial_locations().numpy(), o.coords)     # site lists: exact
        # ~50 layers of fp32 MFMA + fp64-partial BN statistics vs double-accumulated GEMM + sequential-fp32 BN
        # statistics (the reference's arithmetic): 2e-3 of the map's max, measured ~2e-4
        assert _relerr(d.features.detach().cpu().numpy(), o.v) < 2e-3, i
    for d, o in zip(roi_maps, o_roi):
        assert _relerr(d.features.detach().cpu().numpy(), o.v) < 2e-3
    # every BN output agrees, and ReLU masks flip only at rounding distance of 0
    flips = 0
    for name, a in acts.items():
        assert _relerr(a, fo.acts[name]) < 2e-3, name
        flips += int(((a > 0) != (fo.acts[name] > 0)).sum())
    assert flips <= 1e-4 * sum(a.size for a in acts.values()) + 4
    # running statistics (momentum 0.95; unbiased variance) of one deep BN
    nm = "m_downs.4.block0.bn2"
    np.testing.assert_allclose(bn_mods[nm].running_mean.cpu().numpy(), P[nm]["running_mean_out"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(bn_mods[nm].running_var.cpu().numpy(), P[nm]["running_var_out"], rtol=2e-3, atol=2e-5)
    # the reference's multiply-add counter
    import sparseconvnet as scn
    scn.forward_pass_multiplyAdd_count = 0
    with torch.no_grad():
        net.eval()
        net([_t(locs), _t(feats)])
        net.train()
    assert float(scn.forward_pass_multiplyAdd_count) == fo.macs
    # ---- backward: the oracle replays its forward on the DEVICE's BN outputs (identical ReLU masks), which
    # isolates the backward kernels from mask flips at activations within rounding distance of 0
    rng = np.random.default_rng(3)
    G = [rng.standard_normal(m.features.shape).astype(np.float32) / m.features.shape[0] for m in rpn_maps]
    torch.autograd.backward([m.features for m in rpn_maps], [_t(g) for g in G])
    P2 = ref_net.fpn_params(net)
    for k, v in P.items():               # oracle forward again from the ORIGINAL running stats
        if isinstance(v, dict):
            P2[k]["running_mean"], P2[k]["running_var"] = v["running_mean"], v["running_var"]
    fo2 = _oracle_net(P2)
    fo2.override = acts
    fo2.forward(locs, feats)
    grads = fo2.backward(G)
    names = ref_net.fpn_param_names(net)
    checked = 0
    for key, par in names.items():
        if key not in grads:
            assert par.grad is None or float(par.grad.abs().max()) == 0.0, key   # dead branches (ups 5..8)
            continue
        got = par.grad.detach().cpu().numpy().reshape(grads[key].shape)
        assert _relerr(got, grads[key]) < 3e-3, key
        checked += 1
    assert checked >= 100
    assert _relerr(f.grad.cpu().numpy(), grads["d_feats"]) < 3e-3


def _rpn_head_outputs(maps, A, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    objs = [torch.randn(m.features.shape[0] * A, device=DEV, generator=g) for m in maps]
    regs = [torch.randn(m.features.shape[0] * A, 7, device=DEV, generator=g) * 0.2 for m in maps]
    return objs, regs


def _site_counts_by_scale(locs):
    il = O.input_layer(locs, None, 4)
    sites, size, out = il["coords"], np.array(S.FULL_SCALE), {tuple(S.FULL_SCALE): il["V"]}
    for _ in range(8):
        osz = (size - 2) // 2 + 1
        _, sites = O.convolution_rules(sites, [2, 2, 2], [2, 2, 2], osz)
        size = osz
        out[tuple(int(v) for v in size)] = sites.shape[0]
    return out


def test_config2_full_size_bf16_with_nms():
    """BASELINE.json configs[2] at its stated size: 4 x S80k @ 2 cm, bf16 feature storage, whole FPN_Net forward +
    backward, cross-scale proposals with the rotated NMS.  Size-independent properties: per-scale site counts ==
    oracle geometry, bit-reproducible forward, finite gradients, NMS survivors pairwise below the threshold and
    idempotent under a second NMS."""
    import sparseconvnet as scn
    import rpn_glue
    import _nms
    sys.path.insert(0, REPO)
    import bench
    torch.manual_seed(0)
    net = _fpn(feature_dtype=torch.bfloat16).to(DEV)
    net.voxel_scale = 50
    locs, feats = S.make_batch(4, 80000, 9000, 50)
    l, f = _t(locs), _t(feats)
    rpn, roi = net([l, f])
    want = _site_counts_by_scale(locs)
    for m in rpn[:3] + roi:
        assert m.features.shape[0] == want[tuple(m.spatial_size.tolist())]
    assert rpn[0].metadata.input["V"] == want[tuple(S.FULL_SCALE)]
    with torch.no_grad():
        rpn2, _ = net([l, f])
    for a, b in zip(rpn, rpn2):
        assert torch.equal(a.features, b.features)                     # no atomics anywhere in the accumulation
    sum(m.features.square().mean() for m in rpn).backward()
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    base, strides = bench.rpn_constants(torch)
    objs, regs = _rpn_head_outputs(rpn, 4, 1)
    props = rpn_glue.rpn_proposals(rpn, objs, regs, base, strides, 50.0, 2000, 1000, 0.5, (0.3, 0.3))
    assert len(props) == 4
    for boxes, scores in props:
        n = boxes.shape[0]
        assert 0 < n <= 1000 and torch.isfinite(boxes).all()
        assert (scores[:-1] >= scores[1:]).all()                      # descending score order kept
        nb = boxes.clone()
        nb[:, 3:5] = nb[:, 3:5].clamp(min=0.3)
        nb[:, 5] = nb[:, 5].clamp(min=0.3)
        iou = _nms.boxes_iou_3d(nb, nb, (0, 0, 0, 0), -1, True)
        iou.fill_diagonal_(0)
        assert float(iou.max()) < 0.5 + 1e-5                           # survivors do not suppress each other
        again = _nms.rotate_nms_sorted(nb, 0.5, 1000, True)
        assert again.numel() == n                                       # idempotent


def test_config4_full_size_whole_network_bf16():
    """BASELINE.json configs[4] at its stated size: one 1.5 M-point scene @ 2 cm, whole FPN_Net, bf16 feature
    storage, forward + backward: per-scale site counts == oracle geometry, bit-reproducibility, finite grads,
    the multiply-add counter == sum over the oracle's rule books."""
    import sparseconvnet as scn
    torch.manual_seed(0)
    net = _fpn(feature_dtype=torch.bfloat16).to(DEV)
    locs, feats = S.make_batch(1, 1500000, 0, 50)
    l, f = _t(locs), _t(feats)
    scn.forward_pass_multiplyAdd_count = 0
    rpn, roi = net([l, f])
    macs = float(scn.forward_pass_multiplyAdd_count)
    want = _site_counts_by_scale(locs)
    assert rpn[0].metadata.input["V"] == want[tuple(S.FULL_SCALE)] > 800000
    for m in rpn[:3] + roi:
        assert m.features.shape[0] == want[tuple(m.spatial_size.tolist())]
    assert macs > 1e11
    with torch.no_grad():
        rpn2, _ = net([l, f])
    for a, b in zip(rpn, rpn2):
        assert torch.equal(a.features, b.features)
    sum(m.features.square().mean() for m in rpn).backward()
    n_grad = 0
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
            n_grad += 1
    assert n_grad > 100


def test_bench_gpus2_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` must produce a 2-rank run (ADVICE r1 / VERDICT r1 #5).  One GPU here: both ranks share
    cuda:0 and the collective goes through gloo -- the launcher, sharding and all-reduce path are the real ones."""
    env = dict(os.environ, AABR_BENCH_SHARE_GPU="1", AABR_BENCH_BACKEND="gloo", AABR_BENCH_PIN="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batches", "1", "--no-cpu-baseline", "--no-extras", "--min-timed-s", "0.2"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0
    # the N > 1 line says what ran it and how evenly the ranks ran
    d = out["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["allreduce_bytes"] > 80e6
    assert d["allreduce_wait_ms_p50"] is not None and d["rank_ms_per_step_min"] <= d["rank_ms_per_step_max"]
    # the gradient all-reduce runs in buckets, all but the last two launched while backward is still running
    gb = d["grad_buckets"]
    assert gb["buckets"] == 5 and gb["launched_during_backward"] == 3 and gb["bytes"] > 70e6
    assert abs(d["rank_ms_per_step_max"] - out["ms_per_step"]) < 1e-6
    t = out["timing"]
    assert out["steps"] == 3 and t["prewarm_steps"] % 5 == 0 and t["prewarm_steps"] >= 10
    if out["timed_region_s"] < 0.2:      # a region shorter than --min-timed-s is cross-checked by a second, longer one
        assert t["extended"]["steps"] >= 1 and t["extended"]["timed_region_s"] >= 0.15
    assert set(t["step_ms"]) >= {"p50", "p90", "min", "max", "first5"}


def test_fpn_prepare_on_side_stream_is_bit_identical():
    """FPN_Net.prepare builds the next batch's geometry (voxel grid, strided grids, rule tables, block streams) on
    a side stream; the forward that picks it up must give the same bits as the plain path (fp32 and bf16)."""
    for fdt in (torch.float32, torch.bfloat16):
        torch.manual_seed(2)
        net = _fpn(feature_dtype=fdt).to(DEV)
        locs, feats = S.make_batch(2, 20000, 31, 20)
        l = _t(locs)

        def run(prepared):
            f = _t(feats).requires_grad_(True)
            net.zero_grad()
            if prepared:
                side = torch.cuda.Stream()
                net.prepare([l, f], side)
                assert len(net.layers_in[0]._prepared) == 1
            rpn, roi = net([l, f])
            if prepared:
                assert len(net.layers_in[0]._prepared) == 0 and rpn[0].metadata._fpn_prebuilt is not None
            sum(m.features.square().mean() for m in rpn).backward()
            torch.cuda.synchronize()
            return [m.features.detach().clone() for m in rpn], f.grad.clone(), \
                [p.grad.clone() for p in net.parameters() if p.grad is not None]

        a, b = run(False), run(True)
        # BN running statistics moved between the two runs, but training-mode BN uses batch statistics: same bits
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], b[1])
        for x, y in zip(a[2], b[2]):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_geometry_prefetch_thread_is_bit_identical_and_overlaps_training():
    """sparseconvnet.GeometryPrefetcher: the next batch's geometry built by the helper thread (its own stream, its own
    geometry recorder) WHILE the training thread runs forward + backward of the current batch gives the same bits as
    building it inline, over several alternating batches; an exception in the helper surfaces in wait()."""
    torch.manual_seed(5)
    net = _fpn().to(DEV)
    batches = []
    for seed in (41, 43, 47):
        locs, feats = S.make_batch(2, 15000, seed, 20)
        batches.append((_t(locs), _t(feats)))

    def train(l, f):
        f = f.clone().requires_grad_(True)
        net.zero_grad()
        rpn, _ = net([l, f])
        sum(m.features.square().mean() for m in rpn).backward()
        return [m.features.detach().clone() for m in rpn], f.grad.clone()

    order = [0, 1, 2, 1, 0, 2]
    inline = [train(*batches[i]) for i in order]
    torch.cuda.synchronize()
    pf = net.prefetcher()
    assert pf is net.prefetcher()
    got = []
    pf.submit(list(batches[order[0]]))
    for j, i in enumerate(order):
        pf.wait()
        assert len(net.layers_in[0]._prepared) == 1
        l, f = batches[i]
        f2 = f.clone().requires_grad_(True)
        net.zero_grad()
        rpn, _ = net([l, f2])                                   # picks the helper's Metadata up (matched by `l`)
        assert len(net.layers_in[0]._prepared) == 0 and rpn[0].metadata._fpn_prebuilt is not None
        if j + 1 < len(order):
            pf.submit(list(batches[order[j + 1]]))              # built while this batch's backward runs
        sum(m.features.square().mean() for m in rpn).backward()
        got.append(([m.features.detach().clone() for m in rpn], f2.grad.clone()))
    torch.cuda.synchronize()
    for a, b in zip(inline, got):
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], b[1])
    pf.submit([None, None])                                     # not a batch: fails on the host, before any launch
    with pytest.raises(AttributeError):
        pf.wait()
    pf.submit(list(batches[0]))                                 # the helper survives its own exception
    pf.wait()
    net.layers_in[0]._prepared.clear()
    pf.close()


@pytest.mark.gpu
def test_fpn_weights_packed_once_per_version_is_bit_identical():
    """FPN_Net packs every convolution weight (both orientations) with one launch per weight version
    (SCN.WeightPackPlan) instead of one launch per layer call: same bits as the per-call packing, the packs are
    really used, and an in-place weight update is picked up (version check), fp32 and bf16."""
    import sparseconvnet as scn
    from sparseconvnet import SCN
    for fdt in (torch.float32, torch.bfloat16):
        torch.manual_seed(4)
        net = _fpn(feature_dtype=fdt).to(DEV)
        locs, feats = S.make_batch(2, 20000, 37, 20)
        l = _t(locs)

        def run(prepack):
            net.prepack_weights = prepack
            if not prepack:
                for p in net.parameters():
                    if hasattr(p, "_aabr_pack"):
                        del p._aabr_pack
            f = _t(feats).requires_grad_(True)
            net.zero_grad()
            rpn, roi = net([l, f])
            sum(m.features.square().mean() for m in rpn).backward()
            torch.cuda.synchronize()
            return [m.features.detach().clone() for m in rpn], f.grad.clone(), \
                [p.grad.clone() for p in net.parameters() if p.grad is not None]

        a = run(False)
        SCN.pack_stats.update(plan=0, own=0)
        b = run(True)
        assert SCN.pack_stats["own"] == 0 and SCN.pack_stats["plan"] >= 40, SCN.pack_stats
        convs = [m for m in net.modules() if isinstance(m, (scn.SubmanifoldConvolution, scn.Convolution,
                                                             scn.Deconvolution))]
        # the plan covers every convolution, no pack launch of a layer's own ran, nothing is left on the parameters
        assert len(net._pack_plan.weights) == len(convs) and not any(hasattr(m.weight, "_aabr_pack") for m in convs)
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], b[1])
        for x, y in zip(a[2], b[2]):
            assert torch.equal(x, y)
        # an update through .data (no version counter moves -- dp.FlatParams does this): the next forward must
        # see the new weights
        for m in convs:
            m.weight.data.mul_(0.5)
        c = run(True)
        d = run(False)
        assert not torch.equal(b[0][0], c[0][0])
        for x, y in zip(c[0], d[0]):
            assert torch.equal(x, y)
        for x, y in zip(c[2], d[2]):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_compiled_graph_pipelined_hand_over_is_bit_equal_to_one_call():
    """planExecutor.pipeline_records (the pass's list handed to the launcher thread in parts while the next part is
    filled): outputs, input gradient, parameter gradients and running statistics are the bits of the one-call form,
    for part sizes from one record up; fp32 and bf16 storage."""
    from sparseconvnet import planExecutor
    keep = planExecutor.pipeline_records
    try:
        for fdt in (torch.float32, torch.bfloat16):
            torch.manual_seed(8)
            net = _fpn(feature_dtype=fdt).to(DEV)
            net.compiled_graph = True
            state = {k: v.clone() for k, v in net.state_dict().items()}
            locs, feats = S.make_batch(2, 20000, 43, 20)
            l = _t(locs)

            def run(parts):
                planExecutor.pipeline_records = parts
                net.load_state_dict(state)
                net.train(True)
                f = _t(feats).requires_grad_(True)
                net.zero_grad()
                rpn, roi = net([l, f])
                outs = [m.features.detach().clone() for m in rpn + roi]
                sum(m.features.float().square().mean() for m in rpn + roi).backward()
                torch.cuda.synchronize()
                return (outs, f.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                        {k: v.clone() for k, v in net.state_dict().items() if "running" in k})

            ref = run(0)
            for parts in (1, 7, 32, 1000):
                got = run(parts)
                for x, y in zip(ref[0], got[0]):
                    assert torch.equal(x, y)
                assert torch.equal(ref[1], got[1])
                assert ref[2].keys() == got[2].keys()
                for n in ref[2]:
                    assert torch.equal(ref[2][n], got[2][n]), (parts, n)
                for k in ref[3]:
                    assert torch.equal(ref[3][k], got[3][k]), (parts, k)
    finally:
        planExecutor.pipeline_records = keep


def test_prune_unused_levels_is_bit_identical_and_opt_in():
    """`FPN_Net.prune_unused_levels` (off by default -- the reference runs the whole top-down path,
    fpn_net.py:181-196) stops the top-down path behind the last consumed level.  Returned maps, input gradient and the
    gradients of every live parameter are bit-identical to the unpruned pass (module path and compiled graph, both site
    orders); the dead stages' parameters get no gradient either way; the MAC counter is smaller by their launches."""
    import sparseconvnet as scn
    torch.manual_seed(8)
    net = _fpn().to(DEV)
    assert net.prune_unused_levels is False and net._top_down_levels() == 8
    locs, feats = S.make_batch(2, 20000, 43, 20)
    l = _t(locs)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def run(prune, compiled, order):
        net.load_state_dict(state)
        net.prune_unused_levels, net.compiled_graph = prune, compiled
        net.set_site_order(order)
        f = _t(feats).requires_grad_(True)
        net.zero_grad()
        scn.forward_pass_multiplyAdd_count = 0
        rpn, roi = net([l, f])
        macs = float(scn.forward_pass_multiplyAdd_count)
        w = [torch.linspace(0.5, 1.5, m.features.numel(), device=DEV).view_as(m.features) for m in rpn + roi]
        sum((m.features * wi).square().mean() for m, wi in zip(rpn + roi, w)).backward()
        torch.cuda.synchronize()
        return ([m.features.detach().clone() for m in rpn + roi], f.grad.clone(), macs,
                {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0})

    try:
        for compiled in (False, True):
            for order in ("first_seen", "brick"):
                a, b = run(False, compiled, order), run(True, compiled, order)
                assert net._top_down_levels() == 4
                for x, y in zip(a[0], b[0]):
                    assert torch.equal(x, y)
                assert torch.equal(a[1], b[1])
                assert b[2] < a[2]                                    # fewer multiply-adds counted ...
                assert a[3].keys() == b[3].keys()                     # ... and the same parameters receive a gradient
                for n in a[3]:
                    assert torch.equal(a[3][n], b[3][n]), n
                assert not any(n.startswith(("m_ups.4", "m_ups.7", "m_mergeds.4", "m_mergeds.7", "m_shortcuts.0"))
                               for n in a[3])
    finally:
        net.prune_unused_levels = False


def test_fpn_compiled_graph_matches_module_path():
    """FPN_Net.compiled_graph (sparseconvnet/planExecutor.py: every layer between the input layer and the returned
    maps as one launch list per pass, one autograd node) against the module path on the same net and input:
    feature maps, input gradient and BatchNorm running statistics bit-equal; parameter gradients bit-equal where
    at most two gradient contributions meet and within 1e-6 relative otherwise; MAC counter equal; fp32 and bf16;
    evaluation mode; fallback when a hook is registered."""
    import sparseconvnet as scn
    from sparseconvnet import planExecutor
    for fdt in (torch.float32, torch.bfloat16):
        torch.manual_seed(6)
        net = _fpn(feature_dtype=fdt).to(DEV)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        locs, feats = S.make_batch(2, 20000, 41, 20)
        l = _t(locs)

        def run(compiled, train=True):
            net.load_state_dict(state)
            net.train(train)
            net.compiled_graph = compiled
            f = _t(feats).requires_grad_(train)
            net.zero_grad()
            scn.forward_pass_multiplyAdd_count = 0
            before = planExecutor.stats["passes"]
            with torch.set_grad_enabled(train):
                rpn, roi = net([l, f])
            assert (planExecutor.stats["passes"] - before) == (1 if compiled else 0)
            macs = float(scn.forward_pass_multiplyAdd_count)
            outs = [m.features.detach().clone() for m in rpn + roi]
            sizes = [tuple(m.spatial_size.tolist()) for m in rpn + roi]
            if not train:
                return outs, sizes, macs
            w = [torch.linspace(0.5, 1.5, m.features.numel(), device=DEV).view_as(m.features) for m in rpn + roi]
            sum((m.features * wi).square().mean() for m, wi in zip(rpn + roi, w)).backward()
            torch.cuda.synchronize()
            bn = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
            return outs, sizes, macs, f.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()
                                                       if p.grad is not None}, bn

        a, b = run(False), run(True)
        assert a[1] == b[1] and a[2] == b[2] and len(a[0]) == len(b[0])
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[3], b[3])
        assert a[4].keys() == b[4].keys()
        exact = 0
        for n in a[4]:
            ga, gb = a[4][n], b[4][n]
            if torch.equal(ga, gb):
                exact += 1
            else:
                tol = 1e-6 if fdt == torch.float32 else 1e-2
                assert float((ga - gb).abs().max()) <= tol * float(ga.abs().max()), n
        assert exact >= len(a[4]) * 3 // 4, (exact, len(a[4]))
        for k in a[5]:
            assert torch.equal(a[5][k], b[5][k]), k
        # evaluation mode (running statistics), no autograd
        ea, eb = run(False, train=False), run(True, train=False)
        for x, y in zip(ea[0], eb[0]):
            assert torch.equal(x, y)
        # a hook anywhere: the modules run
        h = net.m_mergeds[0].register_forward_hook(lambda mod, i, o: None)
        before = planExecutor.stats["passes"]
        net.compiled_graph = True
        with torch.no_grad():
            net([l, _t(feats)])
        assert planExecutor.stats["passes"] == before
        h.remove()


@pytest.mark.gpu
def test_fpn_compiled_graph_is_bit_reproducible():
    """the compiled graph runs its weight gradients on a second stream and folds adds into their producers: the same
    step twice (fresh geometry each time) must give the same bits -- maps, input gradient, every parameter gradient"""
    torch.manual_seed(8)
    net = _fpn().to(DEV)
    net.compiled_graph = True
    state = {k: v.clone() for k, v in net.state_dict().items()}
    locs, feats = S.make_batch(2, 25000, 43, 20)
    l = _t(locs)

    def run():
        net.load_state_dict(state)
        f = _t(feats).requires_grad_(True)
        net.zero_grad()
        rpn, roi = net([l, f])
        sum(m.features.square().mean() for m in rpn + roi).backward()
        torch.cuda.synchronize()
        return [m.features.detach().clone() for m in rpn], f.grad.clone(), \
            [p.grad.clone() for p in net.parameters() if p.grad is not None]

    a, b, c = run(), run(), run()
    for other in (b, c):
        for x, y in zip(a[0], other[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], other[1])
        assert len(a[2]) == len(other[2]) > 100
        for x, y in zip(a[2], other[2]):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_fpn_compiled_graph_with_partial_output_gradients():
    """only some of the returned maps feed the loss: the compiled graph's backward list is generated for that
    pattern (no launches for branches nothing flows back through) and must agree with autograd over the modules --
    same set of parameters with a gradient, same values"""
    torch.manual_seed(9)
    net = _fpn().to(DEV)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    locs, feats = S.make_batch(2, 20000, 47, 20)
    l = _t(locs)

    def run(compiled):
        net.load_state_dict(state)
        net.compiled_graph = compiled
        f = _t(feats).requires_grad_(True)
        net.zero_grad()
        rpn, roi = net([l, f])
        (rpn[0].features.square().mean() + rpn[4].features.abs().mean()).backward()
        torch.cuda.synchronize()
        return f.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    (fa, ga), (fb, gb) = run(False), run(True)
    assert torch.equal(fa, fb)
    nz_a = {n for n, g in ga.items() if float(g.abs().max()) > 0}
    nz_b = {n for n, g in gb.items() if float(g.abs().max()) > 0}
    assert nz_a == nz_b and 10 < len(nz_b) < len(list(net.parameters()))
    for n in nz_a:
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-6 * float(ga[n].abs().max()), n


def test_bucketed_gradient_allreduce_equals_flat_update():
    """dp.FlatParams.begin_bucketed / finish_bucketed (the compiled backward hands its gradient buffer over in
    slices, each all-reduced in place while backward runs: planExecutor.on_grads_ready) must leave the parameters
    bit-equal to the flat path (gradients packed after backward, one all-reduce, one update).  One rank, gloo: the
    collective is the identity, what is compared is which gradient reaches which parameter."""
    import torch.distributed as dist
    import dp
    from sparseconvnet import planExecutor
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500))
    try:
        torch.manual_seed(11)
        net = _fpn().to(DEV)
        net.compiled_graph = True
        head = torch.nn.Linear(128, 4).to(DEV)
        flat = dp.FlatParams([net, head])
        locs, feats = S.make_batch(2, 15000, 77, 20)
        l, f = _t(locs), _t(feats)
        start = flat.flat.clone()

        def run(bucketed):
            flat.flat.copy_(start)
            for b in net.buffers():
                if b.dtype.is_floating_point:
                    b.zero_() if "mean" in str(b.shape) else None
            flat.zero_grad()
            if bucketed:
                planExecutor.grad_segments = 4
                flat.begin_bucketed()
            rpn, _ = net([l, f])
            loss = sum(head(m.features).square().mean() for m in rpn)
            loss.backward()
            if bucketed:
                n = flat.finish_bucketed(0.5, 1)
                planExecutor.grad_segments = 0
                return n
            flat.sgd_step(0.5, 1)
            return 0

        state = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
        run(False)
        want = flat.flat.clone()
        net.load_state_dict(state, strict=False)
        n = run(True)
        # the first bucketed step only collects its buckets: the plan is agreed across the ranks BEFORE any data
        # collective is launched (dp.py); from the second step on they start underneath the backward pass
        assert n == 5 and flat.bucket_stats["launched_during_backward"] == 0
        assert flat.bucket_stats["bytes"] >= 4 * sum(p.numel() for p in flat.params if p.grad is not None)
        assert torch.equal(flat.flat, want)
        assert planExecutor.on_grads_ready is None
        net.load_state_dict(state, strict=False)
        n = run(True)
        assert n == 5 and flat.bucket_stats["launched_during_backward"] == 3
        assert torch.equal(flat.flat, want)
    finally:
        if own_group:
            dist.destroy_process_group()


def test_rccl_world_size_1_bucketed_equals_flat():
    """the bench's training step under backend `nccl` (RCCL) at world_size 1 with the bucketed
    gradient all-reduce forced on, bit-equal to the flat update -- RCCL init, async work handles and the real stream
    ordering between aabr_plan_run's side-stream join, the hook and the collective (tests/nccl_ws1_child.py, a fresh
    child process; reference: tools/train_net_sparse3d.py:64-69,183-190)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               AABR_BENCH_PIN="0")
    port = str(29700 + os.getpid() % 200)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "nccl_ws1_child.py"), port], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["backend"] == "nccl" and out["world_size"] == 1
    assert out["max_param_change"] > 0                      # the steps really moved the parameters
    assert out["equal"] and out["reproducible"], out
    assert out["buckets"] == 5 and out["launched_during_backward"] == 3 and out["bucket_bytes"] > 70e6
    assert out["hook_disarmed"] and out["grads_are_means"]


def test_inference_pass_packs_its_arena_by_liveness():
    """VERDICT r3 housekeeping: a pass nobody differentiates (torch.no_grad) hands a buffer's bytes on after its last
    reader instead of keeping every activation of the network; same bits as the unpacked pass, a fraction of the bytes"""
    from sparseconvnet import planExecutor
    for fdt in (torch.float32, torch.bfloat16):
        torch.manual_seed(12)
        net = _fpn(feature_dtype=fdt).to(DEV)
        net.compiled_graph = True
        net.eval()
        locs, feats = S.make_batch(2, 30000, 51, 20)
        l, f = _t(locs), _t(feats)
        outs = {}
        for packed in (False, True):
            planExecutor.pack_inference_arena = packed
            planExecutor.stats.pop("arena_bytes_packed", None)
            try:
                with torch.no_grad():
                    rpn, roi = net([l, f])
                    torch.cuda.synchronize()
                    outs[packed] = [m.features.clone() for m in rpn + roi]
            finally:
                planExecutor.pack_inference_arena = True
            if packed:
                pk, fl = planExecutor.stats["arena_bytes_packed"], planExecutor.stats["arena_bytes_flat"]
                assert 0 < pk < 0.35 * fl, (pk, fl)
            else:
                assert "arena_bytes_packed" not in planExecutor.stats
        for a, b in zip(outs[False], outs[True]):
            assert torch.equal(a, b)

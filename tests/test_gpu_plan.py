"""aabr_plan_run and its helpers through the C ABI: elementwise add / storage cast against torch (bit-equal),
rule totals of many rule books in one launch, a two-record plan with a side-stream record and a join, the error
path (unknown record kind), and the strided grids built in rounds from a base grid against the level-by-level
construction."""
import struct

import numpy as np
import pytest
import torch

import synth_scenes as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OP = struct.Struct("<ii6i4f4q12Q")


def _lib():
    import _hip
    return _hip, _hip.load()


def _rec(kind, flags, i32=(), f32=(), i64=(), ps=()):
    return OP.pack(kind, flags, *(tuple(i32) + (0,) * (6 - len(i32))), *(tuple(f32) + (0.0,) * (4 - len(f32))),
                   *(tuple(i64) + (0,) * (4 - len(i64))), *(tuple(ps) + (0,) * (12 - len(ps))))


def test_add_and_cast_match_torch_bitwise():
    _hip, lib = _lib()
    torch.manual_seed(0)
    for n in (1, 3, 4, 1023, 1024, 100003):
        a, b = torch.randn(n, device=DEV), torch.randn(n, device=DEV) * 1e3
        out = torch.empty_like(a)
        _hip.check(lib.aabr_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, 0, _hip.stream()))
        assert torch.equal(out, a + b)
        a16, b16 = a.bfloat16(), b.bfloat16()
        o16 = torch.empty_like(a16)
        _hip.check(lib.aabr_add(a16.data_ptr(), b16.data_ptr(), o16.data_ptr(), n, 1, _hip.stream()))
        assert torch.equal(o16, a16 + b16)
        c16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        _hip.check(lib.aabr_cast_storage(a.data_ptr(), c16.data_ptr(), n, 1, _hip.stream()))
        assert torch.equal(c16, a.to(torch.bfloat16))
        c32 = torch.empty(n, dtype=torch.float32, device=DEV)
        _hip.check(lib.aabr_cast_storage(c16.data_ptr(), c32.data_ptr(), n, 0, _hip.stream()))
        assert torch.equal(c32, c16.float())


def test_sum_counts_many_jobs():
    import ctypes as C
    _hip, lib = _lib()
    rng = np.random.default_rng(1)
    sizes = [0, 1, 255, 256, 257, 40000] + [int(v) for v in rng.integers(1, 5000, 70)]   # > 64 jobs: two launches
    cs = [torch.as_tensor(rng.integers(0, 2000, n).astype(np.int32)).to(DEV) for n in sizes]
    out = torch.full((len(sizes),), -1.0, dtype=torch.float64, device=DEV)
    n = len(sizes)
    _hip.check(lib.aabr_sum_counts((C.c_void_p * n)(*[c.data_ptr() if c.numel() else None for c in cs]),
                                   (C.c_int64 * n)(*sizes),
                                   (C.c_void_p * n)(*[out.data_ptr() + 8 * i for i in range(n)]), n, _hip.stream()))
    assert out.tolist() == [float(c.long().sum().item()) for c in cs]


def test_plan_run_records_side_stream_join_and_error_path():
    _hip, lib = _lib()
    n = 1 << 20
    a, b = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    s1, s2, s3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    plan = b"".join([
        _rec(6, 4, i64=(n,), ps=(a.data_ptr(), b.data_ptr(), s1.data_ptr())),        # s1 = a + b on the second stream
        _rec(6, 0, i64=(n,), ps=(a.data_ptr(), a.data_ptr(), s2.data_ptr())),        # s2 = a + a meanwhile
        _rec(6, 8, i64=(n,), ps=(s1.data_ptr(), s2.data_ptr(), s3.data_ptr())),      # s3 = s1 + s2: joins first
    ])
    for _ in range(3):
        _hip.check(lib.aabr_plan_run(plan, 3, _hip.stream()))
    torch.cuda.synchronize()
    assert torch.equal(s3, (a + b) + (a + a))
    bad = _rec(99, 0)
    assert lib.aabr_plan_run(bad, 1, _hip.stream()) != 0
    assert b"unknown kind" in lib.aabr_last_error()
    # a failing record reports its entry point's message (misaligned operands of aabr_add)
    bad = _rec(6, 0, i64=(8,), ps=(a.data_ptr() + 4, b.data_ptr(), s1.data_ptr()))
    assert lib.aabr_plan_run(bad, 1, _hip.stream()) != 0
    assert b"aligned" in lib.aabr_last_error()


def test_plan_submit_parts_hold_the_side_stream_and_drain_reports_errors():
    """aabr_plan_submit / aabr_plan_drain (the launcher thread): a list handed over in parts -- the first part leaves
    its second-stream record unjoined (hold_side = 1), the last part's JOIN record waits for it -- gives the result of
    the one-call form; an empty closing part joins what is held; a failing part's code and message come out of
    aabr_plan_drain, the parts behind it are dropped, and the launcher is usable afterwards."""
    _hip, lib = _lib()
    n = 1 << 20
    a, b = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    s1, s2, s3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    r1 = _rec(6, 4, i64=(n,), ps=(a.data_ptr(), b.data_ptr(), s1.data_ptr()))         # second stream
    r2 = _rec(6, 0, i64=(n,), ps=(a.data_ptr(), a.data_ptr(), s2.data_ptr()))
    r3 = _rec(6, 8, i64=(n,), ps=(s1.data_ptr(), s2.data_ptr(), s3.data_ptr()))         # joins first
    st = _hip.stream()
    for _ in range(20):
        s3.zero_()
        _hip.check(lib.aabr_plan_submit(r1, 1, st, 1))
        _hip.check(lib.aabr_plan_submit(r2, 1, st, 1))
        _hip.check(lib.aabr_plan_submit(r3, 1, st, 0))
        _hip.check(lib.aabr_plan_drain())
        assert torch.equal(s3, (a + b) + (a + a))          # (torch's launch queues behind the issued parts)
    # held side work joined by an empty closing part: the caller's stream sees s1
    s1.zero_()
    _hip.check(lib.aabr_plan_submit(r1, 1, st, 1))
    _hip.check(lib.aabr_plan_submit(b"", 0, st, 0))
    _hip.check(lib.aabr_plan_drain())
    assert torch.equal(s1 + 0, a + b)
    # error path: the failing part's message, nothing issued behind it
    s2.zero_()
    torch.cuda.synchronize()
    _hip.check(lib.aabr_plan_submit(_rec(99, 0), 1, st, 1))
    _hip.check(lib.aabr_plan_submit(r2, 1, st, 0))
    assert lib.aabr_plan_drain() != 0
    assert b"unknown kind" in lib.aabr_last_error()
    torch.cuda.synchronize()
    assert float(s2.abs().max()) == 0.0
    _hip.check(lib.aabr_plan_drain())                      # the error was consumed
    _hip.check(lib.aabr_plan_submit(r2, 1, st, 0))
    _hip.check(lib.aabr_plan_drain())
    assert torch.equal(s2, a + a)


def test_strided_grids_in_rounds_equal_level_by_level():
    """FPN_Net.grids_from_input (Metadata.buildGridsFromInput: every grid of a round of four levels straight from
    the round's base grid, one host read per round) against the level-by-level construction: every grid's site
    list in the same order, every rule table equal, outputs bit-equal."""
    from test_cabi_and_host import default_fpn
    torch.manual_seed(3)
    net = default_fpn().to(DEV)
    locs, feats = S.make_batch(2, 30000, 77, 20)
    l, f = torch.as_tensor(locs).to(DEV), torch.as_tensor(feats).to(DEV)
    res = []
    for flag in (False, True):
        net.grids_from_input = flag
        with torch.no_grad():
            rpn, roi = net([l, f])
        md = rpn[0].metadata
        grids = {k: g.coords.clone() for k, g in md.grids.items()}
        tabs = {k: (tb.out.table.clone(), tb.inn.table.clone()) for k, tb in md.rulebooks.items()}
        res.append((grids, tabs, [m.features.clone() for m in rpn]))
    (g0, t0, o0), (g1, t1, o1) = res
    assert g0.keys() == g1.keys() and len(g0) >= 13
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert t0.keys() == t1.keys() and len(t0) >= 12
    for k in t0:
        assert torch.equal(t0[k][0], t1[k][0]) and torch.equal(t0[k][1], t1[k][1]), k
    for x, y in zip(o0, o1):
        assert torch.equal(x, y)


def test_grids_carry_per_sample_row_offsets():
    """the per-sample row offsets that ride along with a grid's site-count read (SparseGrid::ctr, Metadata.h:24-33;
    aabr_sample_offsets) equal a bincount of the grid's batch column, for every grid of a 3-sample pass whose
    middle sample is tiny; rpn_proposals then needs no read of its own to slice the maps per example."""
    from test_cabi_and_host import default_fpn
    torch.manual_seed(4)
    net = default_fpn().to(DEV)
    la, fa = S.make_batch(1, 20000, 5, 20)
    lb, fb = S.make_batch(1, 40, 6, 20)
    lc, fc = S.make_batch(1, 9000, 7, 20)
    lb[:, 3], lc[:, 3] = 1, 2
    locs, feats = np.concatenate([la, lb, lc]), np.concatenate([fa, fb, fc])
    with torch.no_grad():
        rpn, _ = net([torch.as_tensor(locs).to(DEV), torch.as_tensor(feats).to(DEV)])
    md = rpn[0].metadata
    n_with = 0
    for k, g in md.grids.items():
        if g.sample_off is None:
            continue
        n_with += 1
        want = torch.bincount(g.coords[:, 3].long(), minlength=3).tolist()
        assert g.sample_counts(3) == want, (k, g.sample_counts(3), want)
        assert g.sample_off[0] == 0 and g.sample_off[3] == g.V and g.sample_counts(5)[3:] == [0, 0]
    assert n_with >= 12
    for m in rpn:   # every RPN map's grid has them
        assert md.grids[tuple(int(v) for v in m.spatial_size.tolist())].sample_off is not None


def test_mailbox_read_back_matches_tolist_also_from_a_side_stream():
    """_hip.read_back (aabr_mailbox_post): values equal tensor.tolist() for the dtypes / shapes the host side reads;
    repeated posts reuse the mailbox (sequence numbers); and a read issued on a side stream while a long queue of
    kernels is pending on the main stream returns the right values (no timing claim: when the posting kernel gets a
    slot beside those kernels is the hardware scheduler's business)."""
    import time
    import _hip
    dev = torch.device(DEV)
    rng = np.random.default_rng(5)
    for dt, shape in ((torch.int32, (7,)), (torch.int64, (3, 5)), (torch.float32, (4,)), (torch.int64, ()),
                      (torch.int32, (6, 2, 3))):
        a = torch.as_tensor(rng.integers(-2 ** 31, 2 ** 31 - 1, size=shape)).to(dt).to(dev)
        for _ in range(3):
            assert _hip.read_back(a) == a.tolist()
    nc = torch.arange(40, device=dev, dtype=torch.int64)[::2]          # non-contiguous view
    assert _hip.read_back(nc) == nc.tolist()
    assert _hip.read_back(torch.zeros(0, dtype=torch.int32, device=dev)) == []
    big = torch.randn(6144, 6144, device=dev)
    side = torch.cuda.Stream()
    small = torch.arange(8, device=dev, dtype=torch.int32)
    big @ big
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    for _ in range(40):
        big @ big                                                      # ~3 ms each on the main stream
    with torch.cuda.stream(side):
        side.wait_event(ev)
        got = _hip.read_back(small * 3)
    torch.cuda.synchronize()
    assert got == [0, 3, 6, 9, 12, 15, 18, 21]


def test_tied_weights_fall_back_to_the_module_path():
    """the compiled backward gives every parameter ONE gradient slot that its single dW launch stores
    into; a weight shared by two layers needs the sum of two contributions.  The template builder refuses such a
    network (planExecutor.Unsupported -> module path), and the shared weight's gradient is the sum."""
    import synth_scenes as S
    import sparseconvnet as scn
    from sparseconvnet import planExecutor
    from test_cabi_and_host import default_fpn
    torch.manual_seed(3)
    net = default_fpn().to(DEV)
    convs = [m for m in net.modules() if isinstance(m, scn.SubmanifoldConvolution) and m.nIn == m.nOut and m.filter_volume == 27]
    a = convs[0]
    b = next(m for m in convs[1:] if m.nIn == a.nIn)
    b.weight = a.weight                                      # tied
    locs, feats = S.make_batch(1, 8000, 3, 20)
    l = torch.as_tensor(locs).to(DEV)

    def run(compiled):
        net.compiled_graph = compiled
        net.zero_grad()
        f = torch.as_tensor(feats).to(DEV).requires_grad_(True)
        rpn, _ = net([l, f])
        sum(m.features.square().mean() for m in rpn).backward()
        return a.weight.grad.clone(), [m.features.detach().clone() for m in rpn]

    before = planExecutor.stats["fallbacks"]
    g_graph, maps_graph = run(True)
    assert planExecutor.stats["fallbacks"] == before + 1
    g_mod, maps_mod = run(False)
    assert torch.equal(g_graph, g_mod)
    for x, y in zip(maps_graph, maps_mod):
        assert torch.equal(x, y)
    assert g_mod.abs().max().item() > 0


def test_parameter_update_between_forward_and_backward_is_refused():
    """the compiled backward reads weight packs and BatchNorm coefficients by reference; an optimizer
    step between forward and backward would silently give gradients of mixed weight versions.  It raises instead
    (as torch does for saved tensors); the standard forward -> backward -> step order is unaffected."""
    import synth_scenes as S
    from test_cabi_and_host import default_fpn
    torch.manual_seed(3)
    net = default_fpn().to(DEV)
    locs, feats = S.make_batch(1, 8000, 3, 20)
    l = torch.as_tensor(locs).to(DEV)
    f = torch.as_tensor(feats).to(DEV).requires_grad_(True)
    rpn, _ = net([l, f])
    loss = sum(m.features.square().mean() for m in rpn)
    with torch.no_grad():
        for p in net.m_downs.parameters():                   # in-place update of the weights inside the compiled graph
            p.add_(1e-3)
    with pytest.raises(RuntimeError, match="modified (in place|by an inplace operation)"):   # ours, or torch's own check
        loss.backward()
    net.zero_grad()
    rpn, _ = net([l, f])
    sum(m.features.square().mean() for m in rpn).backward()  # the regular order works
    assert sum(p.grad is not None for p in net.parameters()) > 100


def test_runs_of_cast_records_go_out_as_one_launch_with_the_same_bits():
    """round 6: consecutive storage-cast records of one direction on the caller's stream are served by ONE job-list launch
    (csrc/plan.hip plan_cast_run; up to 16 records per launch): 19 casts fp32 -> bf16 of odd sizes followed by 5 back,
    against torch's conversion, bit for bit; a lone record keeps the single-launch path."""
    _hip, lib = _lib()
    torch.manual_seed(3)
    K_CAST, F_TO_BF16 = 7, 2
    sizes = [1, 3, 4, 5, 1023, 1024, 1025, 4097, 100003, 7, 8, 9, 2048, 2047, 31, 33, 65536, 12345, 2]
    src = [torch.randn(n, device=DEV) * 10 ** (i % 5 - 2) for i, n in enumerate(sizes)]
    dst = [torch.empty(n, dtype=torch.bfloat16, device=DEV) for n in sizes]
    back = [torch.empty(n, dtype=torch.float32, device=DEV) for n in sizes[:5]]
    recs = b"".join(_rec(K_CAST, F_TO_BF16, i64=(s.numel(),), ps=(s.data_ptr(), d.data_ptr())) for s, d in zip(src, dst))
    recs += b"".join(_rec(K_CAST, 0, i64=(d.numel(),), ps=(d.data_ptr(), b.data_ptr())) for d, b in zip(dst, back))
    _hip.check(lib.aabr_plan_run(recs, len(sizes) + 5, _hip.stream()))
    for s, d in zip(src, dst):
        assert torch.equal(d, s.to(torch.bfloat16))
    for d, b in zip(dst, back):
        assert torch.equal(b, d.float())
    # the records of a list mean "one after the other": a cast whose OUTPUT reuses the memory of an earlier cast's INPUT
    # (what a liveness-packed inference arena does) must not share a launch with it
    n = 50000
    a1, a2 = torch.randn(n, device=DEV), torch.randn(n // 2, device=DEV)
    a1_orig = a1.clone()
    b1 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    b2 = a1.view(torch.bfloat16)[:n // 2]                     # lives in a1's bytes
    recs = _rec(K_CAST, F_TO_BF16, i64=(n,), ps=(a1.data_ptr(), b1.data_ptr())) + \
        _rec(K_CAST, F_TO_BF16, i64=(n // 2,), ps=(a2.data_ptr(), b2.data_ptr()))
    _hip.check(lib.aabr_plan_run(recs, 2, _hip.stream()))
    assert torch.equal(b1, a1_orig.to(torch.bfloat16)) and torch.equal(b2, a2.to(torch.bfloat16))
    one = torch.empty(sizes[8], dtype=torch.bfloat16, device=DEV)
    _hip.check(lib.aabr_plan_run(_rec(K_CAST, F_TO_BF16, i64=(sizes[8],), ps=(src[8].data_ptr(), one.data_ptr())), 1,
                                 _hip.stream()))
    assert torch.equal(one, dst[8])


def test_job_list_stream_builders_equal_book_by_book_builds():
    """round 6: inside a geometry list the tile-block, wide-block and pair-list builders of ALL its rule books go out as one
    launch per kind (csrc/common.h StreamJobs; aabr_geom_run defers them behind the list's last record).  Every stream must
    be word for word what the single-book entry points build: three submanifold books of different sizes and a strided book
    (both sides), wide blocks at two tile sizes, over hash grids and over brick grids."""
    import sparseconvnet as scn
    from sparseconvnet import SCN
    for order in ("first_seen", "brick"):
        tables = []
        for seed, npts in ((5, 3000), (6, 20000), (7, 60000)):
            locs, feats = S.make_batch(2, npts, seed, 20)
            layer = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)
            layer.site_order = order
            x = layer([torch.as_tensor(locs).to(DEV), torch.as_tensor(feats).to(DEV)])
            md = x.metadata
            tb = md.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
            tables.append(tb.out)
            if seed == 7:
                osz = (x.spatial_size - 2) // 2 + 1
                st = md.getRuleBook(x.spatial_size, osz, torch.LongTensor([2, 2, 2]), torch.LongTensor([2, 2, 2]))
                tables += [st.out, st.inn]
        SCN.flush_geom()

        def fresh(g):
            return SCN._Gather(g.table, g._ensure_counts(), g.vol, g.rows)

        def meaningful(g, words, kind, T=0):
            """the words a builder defines (a stream's buffer is sized for the worst case; the tail of a tile's entry
            region, and of an offset's pair list, is never written or read)"""
            w = words.cpu().numpy()
            V, vol = g.rows, g.vol
            if kind == "wide":
                nt, maxb = (V + T - 1) // T, (T // 16) * vol
                pre = w[:nt * (vol + 1)].reshape(nt, vol + 1)
                ent = w[nt * (vol + 1):nt * (vol + 1) + nt * maxb * 16].reshape(nt, maxb * 16)
                return pre, np.where(np.arange(maxb * 16)[None] < pre[:, vol:vol + 1] * 16, ent, 0)
            if kind == "tile":
                nt, maxb = (V + 63) // 64, 4 * vol
                nblk = w[:nt]
                bk = w[nt:nt + nt * maxb].reshape(nt, maxb)
                ent = w[nt + nt * maxb:nt + nt * maxb + nt * maxb * 16].reshape(nt, maxb * 16)
                return (nblk, np.where(np.arange(maxb)[None] < nblk[:, None], bk, 0),
                        np.where(np.arange(maxb * 16)[None] < nblk[:, None] * 16, ent, 0))
            hdr, nb = vol + 2 * (vol + 1), (V + 255) // 256
            pr = w[hdr + vol * nb:hdr + vol * nb + 2 * vol * V].reshape(vol, V, 2)
            return w[:hdr + vol * nb], np.where(np.arange(V)[None, :, None] < w[:vol][:, None, None], pr, 0)

        def streams(gs):
            return [[meaningful(g, g.blocks(), "tile"), meaningful(g, g.blocks_wide(128), "wide", 128),
                     meaningful(g, g.blocks_wide(64), "wide", 64), meaningful(g, g.pairs(), "pairs")] for g in gs]

        want = streams([fresh(g) for g in tables])       # outside a plan: one launch (a one-job list) per call
        listed = [fresh(g) for g in tables]
        with SCN.geom_plan():
            for g in listed:                             # recorded: the list's builders go out as one launch per kind
                g.blocks(), g.blocks_wide(128), g.blocks_wide(64), g.pairs()
        got = streams(listed)                            # (cached now: nothing is rebuilt)
        for g, w_, g_ in zip(tables, want, got):
            assert g.rows > 0
            for a, b in zip(w_, g_):
                for x_, y_ in zip(a, b):
                    assert x_.shape == y_.shape and (x_ == y_).all(), (order, g.rows, g.vol)

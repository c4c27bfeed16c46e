"""Compiled execution of the static part of a SparseConvNet module graph (extension; the reference runs every
layer through its own Python module, autograd Function and pybind call: submanifoldConvolution.py:27-95,
convolution.py:29-90, deconvolution.py:16-87, batchNormalization.py:29-108, tables.py:27-41).

Between the input layer and the returned feature maps FPN_Net (fpn_net.py:168-203) is a fixed dataflow graph of
BatchNorm+LeakyReLU, submanifold / strided / transposed convolutions and residual adds over geometry that is
known before the first feature row is computed.  `run_fpn` walks the SAME module objects once per pass with
symbolic tensors (sizes, no data), turns every layer into one `AabrPlanOp` record of the library's own entry
point -- the launch `SCN.py` would have made, same kernel, same arguments, same order -- and hands the list to
`aabr_plan_run` in one call.  Activations live in one arena; the backward pass is generated the same way in
reverse (input gradients, weight gradients, BatchNorm backward, gradient sums for tensors with several
consumers in the order autograd would add them) and runs as a second list.  One autograd node stands for the
whole graph; parameters, running statistics and state_dict are the modules' own.

What it buys: the ~25 us of interpreter + autograd work per layer and direction leave the critical path (the
4-scene training step of bench.py is host-bound without it in bf16).  What it does not do: change any number --
`tests/test_gpu_fpn.py` holds it bit-equal to the module path.  Falls back (returns None) when a module of the
graph carries hooks or is of a type it does not know."""
import struct

import torch
from torch.autograd import Function

import _hip
from _hip import ptr, stream, check
from . import SCN
from .sparseConvNetTensor import SparseConvNetTensor
from .sequential import Sequential
from .tables import AddTable, ConcatTable
from .identity import Identity
from .batchNormalization import BatchNormalization
from .submanifoldConvolution import SubmanifoldConvolution
from .convolution import Convolution
from .deconvolution import Deconvolution

_OP = struct.Struct("<ii6i4f4q12Q")          # AabrPlanOp (include/aabr_hip.h)
assert _OP.size == 176
K_CONV, K_WIDE, K_DW, K_BNF, K_BNB, K_ADD, K_CAST = 1, 2, 3, 4, 5, 6, 7
F_BF16, F_TO_BF16 = 1, 2
_ALIGN = 256
_Z6, _Z4, _Z12 = (0,) * 6, (0,) * 4, ((0, 0),) * 12
ABS, FWD, GRAD, STAT, PGRAD = 0, 1, 2, 3, 4   # pointer spaces: absolute / arenas (resolved when the arena exists)

stats = {"passes": 0, "fallbacks": 0}


class Unsupported(Exception):
    pass


def _rec(kind, flags, i32=(), f32=(), i64=(), ps=()):
    return (kind, flags, tuple(i32) + _Z6[len(i32):], tuple(f32) + _Z4[len(f32):], tuple(i64) + _Z4[len(i64):],
            tuple(ps) + _Z12[len(ps):])


def _run(recs, bases):
    if not recs:
        return
    buf = bytearray(len(recs) * 176)
    off = 0
    pack = _OP.pack_into
    for kind, flags, i32, f32, i64, ps in recs:
        pack(buf, off, kind, flags, *i32, *f32, *i64, *[bases[s] + o for s, o in ps])
        off += 176
    check(_hip.load().aabr_plan_run(bytes(buf), len(recs), stream()))


class _Bufs(object):
    """bump allocation of [rows, planes] matrices in one arena (offsets first, memory when the total is known)"""

    def __init__(self, space):
        self.space, self.meta, self.total = space, [], 0

    def new(self, rows, planes, dtype):
        nbytes = rows * planes * (2 if dtype == torch.bfloat16 else 4)
        self.meta.append((rows, planes, dtype, (self.space, self.total)))
        self.total += (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        return len(self.meta) - 1


def _p(t):
    """absolute pointer of a live tensor (None -> NULL)"""
    return (ABS, ptr(t) or 0) if t is not None else (ABS, 0)


class _Pass(object):
    """one forward pass through the graph (and the state its backward pass needs)"""

    def __init__(self, net, metadata, dtype, x, x_spatial, train):
        self.net, self.md, self.dtype, self.train = net, metadata, dtype, train
        self.lib = _hip.load()
        self.dev = x.device
        self.bf = F_BF16 if dtype == torch.bfloat16 else 0
        self.a = _Bufs(FWD)
        self.ext = {}                      # forward buffer id -> external tensor (the graph input)
        self.fops = []                     # symbolic forward ops, in the order the module path creates them
        self.frecs = []
        self.params = []                   # parameters in first-use order
        self.pidx = {}
        self.stat_floats = 0
        self.dw_floats = 0
        self.bn_floats = 0
        self.macs, self.hidden = [], 0
        self.x = x
        b = self.a.new(x.size(0), x.size(1), x.dtype)
        self.a.meta[b] = self.a.meta[b][:3] + (_p(x),)
        self.a.total = 0                   # the input is not in the arena
        self.ext[b] = x
        self.x_sym = (b, x_spatial)

    # ---- parameters --------------------------------------------------------------------------------------------
    def param(self, p):
        i = self.pidx.get(id(p))
        if i is None:
            i = self.pidx[id(p)] = len(self.params)
            self.params.append(p)
        return i

    # ---- symbolic forward: one handler per module type ----------------------------------------------------------
    def run(self, m, x):
        t = type(m)
        if isinstance(m, ConcatTable):
            return [self.run(c, x) for c in m._modules.values()]
        if isinstance(m, AddTable):
            return self.add(x)
        if isinstance(m, Sequential):
            for c in m._modules.values():
                x = self.run(c, x)
            return x
        if isinstance(m, Identity):
            return x
        if isinstance(m, BatchNormalization):
            return self.bn(m, x)
        if isinstance(m, SubmanifoldConvolution):
            return self.subm(m, x)
        if isinstance(m, Convolution):
            return self.conv(m, x)
        if isinstance(m, Deconvolution):
            return self.deconv(m, x)
        raise Unsupported(t.__name__)

    def add(self, xs):
        """left-to-right sum (utils._sum_features)"""
        acc = xs[0]
        for t in xs[1:]:
            rows, planes, dt, _ = self.a.meta[acc[0]]
            assert self.a.meta[t[0]][:3] == (rows, planes, dt) and t[1] == acc[1]
            y = self.a.new(rows, planes, dt)
            self.fops.append(("add", acc[0], t[0], y))
            a_, b_, y_ = self.a.meta[acc[0]][3], self.a.meta[t[0]][3], self.a.meta[y][3]
            self.frecs.append(_rec(K_ADD, self.bf if dt == torch.bfloat16 else 0, i64=(rows * planes,),
                                   ps=(a_, b_, y_)))
            acc = (y, acc[1])
        return acc

    def cast(self, x, dtype):
        rows, planes, dt, px = self.a.meta[x[0]]
        if dt == dtype:
            return x
        y = self.a.new(rows, planes, dtype)
        self.fops.append(("cast", x[0], y))
        self.frecs.append(_rec(K_CAST, F_TO_BF16 if dtype == torch.bfloat16 else 0, i64=(rows * planes,),
                               ps=(px, self.a.meta[y][3])))
        return (y, x[1])

    def bn(self, m, x):
        if (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or
                not (m.training or m.track_running_stats)):
            raise Unsupported("BatchNormalization with hooks / batch statistics in evaluation mode")
        rows, planes, dt, px = self.a.meta[x[0]]
        assert planes == m.nPlanes, (planes, m.nPlanes)
        y = self.a.new(rows, planes, dt)
        st = self.stat_floats
        self.stat_floats += 2 * planes
        w = m.weight if m.affine else None
        b = m.bias if m.affine else None
        if m.affine:
            self.param(w)
            self.param(b)
        self.bn_floats = max(self.bn_floats, int(self.lib.aabr_bn_scratch_floats(planes)))
        self.fops.append(("bn", x[0], y, m, st))
        if rows:
            self.frecs.append(_rec(K_BNF, self.bf if dt == torch.bfloat16 else 0,
                                   i32=(planes, 1 if m.training else 0), f32=(m.eps, m.momentum, m.leakiness),
                                   i64=(rows,),
                                   ps=(px, self.a.meta[y][3], (STAT, st * 4), (STAT, (st + planes) * 4),
                                       _p(m.running_mean), _p(m.running_var), _p(w), _p(b), ("bn", 0))))
        return (y, x[1])

    def _conv_rec(self, src, rows_in, n_in, dst, rows_out, n_out, gather, weight, wpack, flags, dt):
        """the launch SCN._conv_fwd makes for a prepacked weight"""
        if rows_out == 0:
            return None
        bf = dt == torch.bfloat16
        T = 0 if bf else self.lib.aabr_conv_wide_tile_rows(n_in, n_out, rows_in, rows_out, gather.vol)
        if T:
            return _rec(K_WIDE, 0, i32=(n_in, n_out, gather.vol, flags & 3, T), i64=(rows_in, rows_out),
                        ps=(src, dst, _p(gather.blocks_wide(T)), (ABS, 0), (ABS, 0), _p(wpack)))
        return _rec(K_CONV, F_BF16 if bf else 0, i32=(n_in, n_out, gather.vol, flags | 4), i64=(rows_in, rows_out),
                    ps=(src, dst, _p(gather.blocks()), _p(weight), (ABS, 0), _p(wpack)))

    def _conv_common(self, m, x, out_spatial, tb, g_fwd, rows_out, g_din, din_flags, g_dw):
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            raise Unsupported("convolution with hooks")
        if hasattr(m, "bias") or getattr(m, "groups", 1) != 1:
            raise Unsupported("convolution with bias / groups")
        rows, planes, dt, px = self.a.meta[x[0]]
        assert planes == m.nIn, (planes, m.nIn)
        w = m.weight
        pk = getattr(w, "_aabr_pack", None)
        if pk is None or pk[1] != dt or pk[0] != w._version:
            raise Unsupported("weight packs missing (FPN_Net.prepack_weights)")
        y = self.a.new(rows_out, m.nOut, dt)
        self.param(w)
        self.macs.append((tb, w))
        self.hidden += rows_out * m.nOut
        self.fops.append(("conv", x[0], y, m, g_din, din_flags, g_dw, pk[3]))
        r = self._conv_rec(px, rows, m.nIn, self.a.meta[y][3], rows_out, m.nOut, g_fwd, w, pk[2], 0, dt)
        if r is not None:
            self.frecs.append(r)
        return (y, out_spatial)

    def subm(self, m, x):
        tb = self.md.getSubmanifoldRuleBook(x[1], m.filter_size)
        return self._conv_common(m, x, x[1], tb, tb.out, tb.V_out, tb.out, 1 | 2, tb.out)

    def conv(self, m, x):
        fs, st = SCN._key(m.filter_size), SCN._key(m.filter_stride)
        osz = tuple((i - f) // s + 1 for i, f, s in zip(x[1], fs, st))
        assert all((o - 1) * s + f == i for o, s, f, i in zip(osz, st, fs, x[1])), (x[1], osz, fs, st)
        tb = self.md.getRuleBook(x[1], osz, m.filter_size, m.filter_stride)
        return self._conv_common(m, x, osz, tb, tb.out, tb.V_out, tb.inn, 1, tb.out)

    def deconv(self, m, x):
        fs, st = SCN._key(m.filter_size), SCN._key(m.filter_stride)
        osz = tuple((i - 1) * s + f for i, f, s in zip(x[1], fs, st))
        tb = self.md.getRuleBook(osz, x[1], m.filter_size, m.filter_stride)
        return self._conv_common(m, x, osz, tb, tb.inn, tb.V_in, tb.out, 1, tb.inn)

    # ---- execution ---------------------------------------------------------------------------------------------
    def _workspaces(self):
        ws = {}
        if self.bn_floats:
            ws["bn"] = ptr(_hip.workspace("bn", self.bn_floats, torch.float32, self.dev))
        if self.dw_floats:
            ws["dw"] = ptr(_hip.workspace("dw", self.dw_floats, torch.float32, self.dev))
        return ws

    @staticmethod
    def _resolve(recs, ws):
        """workspace placeholders ("bn", 0) -> absolute pointers"""
        out = []
        for kind, flags, i32, f32, i64, ps in recs:
            if kind in (K_BNF, K_BNB, K_DW):
                ps = tuple((ABS, ws[s]) if isinstance(s, str) else (s, o) for s, o in ps)
            out.append((kind, flags, i32, f32, i64, ps))
        return out

    def forward(self, outs):
        """run the recorded forward launches; returns one tensor (an arena view) per entry of `outs`"""
        self.arena = torch.empty(max(self.a.total, 1), dtype=torch.uint8, device=self.dev)
        self.stat = torch.empty(max(self.stat_floats, 1), dtype=torch.float32, device=self.dev)
        self.bases = {ABS: 0, FWD: self.arena.data_ptr(), STAT: self.stat.data_ptr()}
        _run(self._resolve(self.frecs, self._workspaces()), self.bases)
        self.frecs = None
        res = []
        for b, _ in outs:
            rows, planes, dt, (space, off) = self.a.meta[b]
            if b in self.ext:
                res.append(self.ext[b].view(rows, planes))
                continue
            es = 2 if dt == torch.bfloat16 else 4
            res.append(self.arena[off:off + rows * planes * es].view(dt).view(rows, planes))
        self.outs = [b for b, _ in outs]
        return res

    def backward(self, gouts, need_dx):
        """generate and run the backward launches; returns (d_x or None, [parameter gradient or None])"""
        g = _Bufs(GRAD)
        recs = []
        contrib = {}
        keep = []
        for b, go in zip(self.outs, gouts):
            if go is None:
                continue
            go = go.contiguous()
            keep.append(go)
            rows, planes, dt, _ = self.a.meta[b]
            assert go.dtype == dt and go.numel() == rows * planes
            k = g.new(rows, planes, dt)
            g.meta[k] = g.meta[k][:3] + (_p(go),)
            contrib.setdefault(b, []).append(k)
        g.total = 0                        # external gradients are not in the arena
        poff, ptotal, pgrad = {}, 0, [None] * len(self.params)

        def pslot(p):
            nonlocal ptotal
            i = self.pidx[id(p)]
            if i not in poff:
                poff[i] = ptotal
                ptotal += (p.numel() + 63) // 64 * 64
            return (PGRAD, poff[i] * 4)

        def total(b):
            """sum of the gradient contributions of forward buffer b, in arrival order (autograd's order: the
            consumer created last delivers first); None when nothing flows back"""
            c = contrib.get(b)
            if not c:
                return None
            acc = c[0]
            for nxt in c[1:]:
                rows, planes, dt, _ = g.meta[acc]
                s = g.new(rows, planes, dt)
                recs.append(_rec(K_ADD, F_BF16 if dt == torch.bfloat16 else 0, i64=(rows * planes,),
                                 ps=(g.meta[acc][3], g.meta[nxt][3], g.meta[s][3])))
                acc = s
            return acc

        dx = None
        x0 = self.x_sym[0]
        for op in reversed(self.fops):
            kind = op[0]
            if kind == "add":
                _, a_, b_, y = op
                gy = total(y)
                if gy is not None:       # same order as autograd's AddBackward: first operand, then second
                    contrib.setdefault(a_, []).append(gy)
                    contrib.setdefault(b_, []).append(gy)
            elif kind == "cast":
                _, x, y = op
                gy = total(y)
                if gy is None:
                    continue
                rows, planes, dt, px = self.a.meta[x]
                gx = g.new(rows, planes, dt)
                recs.append(_rec(K_CAST, F_TO_BF16 if dt == torch.bfloat16 else 0, i64=(rows * planes,),
                                 ps=(g.meta[gy][3], g.meta[gx][3])))
                contrib.setdefault(x, []).append(gx)
            elif kind == "bn":
                _, x, y, m, st = op
                gy = total(y)
                if gy is None:
                    continue
                rows, planes, dt, px = self.a.meta[x]
                gx = g.new(rows, planes, dt)
                w = m.weight if m.affine else None
                pw = pslot(m.weight) if m.affine and m.weight.requires_grad else (ABS, 0)
                pb = pslot(m.bias) if m.affine and m.bias.requires_grad else (ABS, 0)
                if rows:
                    recs.append(_rec(K_BNB, F_BF16 if dt == torch.bfloat16 else 0, i32=(planes,),
                                     f32=(0.0, 0.0, m.leakiness), i64=(rows,),
                                     ps=(px, g.meta[gx][3], self.a.meta[y][3], g.meta[gy][3], (STAT, st * 4),
                                         (STAT, (st + planes) * 4), _p(w), pw, pb, ("bn", 0))))
                contrib.setdefault(x, []).append(gx)
            else:
                _, x, y, m, g_din, din_flags, g_dw, wpack_t = op
                gy = total(y)
                if gy is None:
                    continue
                rows, planes, dt, px = self.a.meta[x]
                rows_y = self.a.meta[y][0]
                if x != x0 or need_dx:
                    gx = g.new(rows, planes, dt)
                    r = self._conv_rec(g.meta[gy][3], rows_y, m.nOut, g.meta[gx][3], rows, m.nIn, g_din, m.weight,
                                       wpack_t, din_flags, dt)
                    if r is not None:
                        recs.append(r)
                    contrib.setdefault(x, []).append(gx)
                if m.weight.requires_grad:
                    mc = g_dw.max_chunks(m.nIn, m.nOut)
                    self.dw_floats = max(self.dw_floats, int(self.lib.aabr_conv_dw_scratch_floats(mc, m.nIn, m.nOut)))
                    recs.append(_rec(K_DW, F_BF16 if dt == torch.bfloat16 else 0, i32=(m.nIn, m.nOut, g_dw.vol),
                                     i64=(rows_y, mc),
                                     ps=(px, g.meta[gy][3], _p(g_dw.pairs()), pslot(m.weight), (ABS, 0), ("dw", 0))))
        gx0 = total(x0) if need_dx else None
        garena = torch.empty(max(g.total, 1), dtype=torch.uint8, device=self.dev)
        gparams = torch.empty(max(ptotal, 1), dtype=torch.float32, device=self.dev)
        bases = dict(self.bases)
        bases[GRAD] = garena.data_ptr()
        bases[PGRAD] = gparams.data_ptr()
        _run(self._resolve(recs, self._workspaces()), bases)
        for i, o in poff.items():
            p = self.params[i]
            pgrad[i] = gparams[o:o + p.numel()].view_as(p)
        if gx0 is not None:
            rows, planes, dt, (space, off) = g.meta[gx0]
            es = 2 if dt == torch.bfloat16 else 4
            if space == GRAD:
                dx = garena[off:off + rows * planes * es].view(dt).view(rows, planes)
            else:   # the graph input reached an output untouched: its gradient is the incoming one
                dx = [t for t in keep if t.data_ptr() == off][0]
        return dx, pgrad


class _GraphFunction(Function):
    @staticmethod
    def forward(ctx, ps, outs, x, *params):
        res = ps.forward(outs)
        ctx.ps = ps
        ctx.need_dx = x.requires_grad
        return tuple(res)

    @staticmethod
    def backward(ctx, *gouts):
        ps, ctx.ps = ctx.ps, None
        dx, pgrad = ps.backward(gouts, ctx.need_dx)
        return (None, None, dx) + tuple(pgrad)


def _fpn_graph(net, ps, x):
    """forward_fpn (fpn_net.py:168-203), symbolically, in the module path's order"""
    n = len(net.m_downs)
    downs = []
    for m in net.m_downs:
        x = ps.run(m, x)
        downs.append(x)
    x = ps.run(net.m_shortcuts[-1], x)
    ups = [x]
    for k in range(n - 1):
        j = n - 1 - k - 1
        x = ps.run(net.m_ups[k], x)
        sc = ps.run(net.m_shortcuts[j], downs[j])
        x = ps.add([x, sc])
        ups.append(ps.run(net.m_mergeds[k], x))
    rpn3d = [ups[i] for i in net.fpn_scales_from_top]
    rpn2d = [ps.run(net.convs_pro2d[i], rpn3d[i]) for i in range(len(rpn3d))]
    maps = rpn3d + rpn2d
    rpn = [maps[i] for i in net.rpn_3d_2d_selector]
    roi = [ups[i] for i in net.roi_scales_from_top]
    for i in range(len(rpn3d)):
        assert tuple(int(v) for v in net.rpn_map_sizes[i]) == rpn3d[i][1], (rpn3d[i][1], net.rpn_map_sizes[i])
    return rpn, roi


def run_fpn(net, net1):
    """FPN_Net.forward after layers_in: (rpn_maps, roi_maps) as SparseConvNetTensors, or None when the graph
    holds something the executor does not handle (the caller then runs the modules)."""
    x = net1.features
    if x.dim() != 2 or x.size(0) == 0 or x.dtype != torch.float32:
        return None
    for m in net.modules():        # any hook anywhere in the graph: the modules themselves must run
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            stats["fallbacks"] += 1
            return None
    train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in net.parameters()))
    try:
        ps = _Pass(net, net1.metadata, net.feature_dtype, x, SCN._key(net1.spatial_size), train)
        xs = ps.cast(ps.x_sym, net.feature_dtype)
        rpn, roi = _fpn_graph(net, ps, xs)
        rpn = [ps.cast(t, torch.float32) for t in rpn]
        roi = [ps.cast(t, torch.float32) for t in roi]
    except Unsupported:
        stats["fallbacks"] += 1
        return None
    outs = rpn + roi
    import sparseconvnet
    for tb, w in ps.macs:
        sparseconvnet.forward_pass_multiplyAdd_count += SCN._macs(tb, w)
    sparseconvnet.forward_pass_hidden_states += ps.hidden
    ps.macs = None
    if train:
        feats = _GraphFunction.apply(ps, outs, x, *ps.params)
    else:
        feats = ps.forward(outs)
        ps.fops = None
    stats["passes"] += 1
    res = []
    for (b, sp), f in zip(outs, feats):
        t = SparseConvNetTensor()
        t.metadata = net1.metadata
        t.spatial_size = torch.LongTensor(list(sp))
        t.features = f
        res.append(t)
    return res[:len(rpn)], res[len(rpn):]

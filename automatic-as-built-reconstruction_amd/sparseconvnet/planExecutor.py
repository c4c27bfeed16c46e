"""Compiled execution of the static part of a SparseConvNet module graph (extension; the reference runs every
layer through its own Python module, autograd Function and pybind call: submanifoldConvolution.py:27-95,
convolution.py:29-90, deconvolution.py:16-87, batchNormalization.py:29-108, tables.py:27-41).

Between the input layer and the returned feature maps FPN_Net (fpn_net.py:168-203) is a fixed dataflow graph of
BatchNorm+LeakyReLU, submanifold / strided / transposed convolutions and residual adds over geometry that is
known before the first feature row is computed.  `run_fpn` turns every layer into one `AabrPlanOp` record of the
library's own entry point -- the launch `SCN.py` would have made, same kernel, same arguments, same order -- and
hands the list to `aabr_plan_run` in one call.  Activations live in one arena; the backward pass is a second list
generated in reverse (input gradients, weight gradients, BatchNorm backward, gradient sums for tensors with
several consumers in the order autograd would add them).  One autograd node stands for the whole graph;
parameters, running statistics and state_dict are the modules' own.

Two stages, so that a pass costs little Python:
  * `_Template` (once per network / input size / dtype / mode): walks the SAME module objects with symbolic
    tensors and records the structure -- which buffer feeds which launch, plane counts, rule-book keys, parameter
    and weight-pack addresses, the backward list and its gradient buffers;
  * a pass fills in what the scene decides: rows per scale, rule-book streams, arena addresses.

What it buys: the ~25 us of interpreter + autograd work per layer and direction leave the critical path.  What it
does not do: change any number -- `tests/test_gpu_fpn.py` holds it bit-equal to the module path.  Falls back
(returns None) when a module of the graph carries hooks or is of a type it does not know."""
import struct

import torch
from torch.autograd import Function

import _hip
from _hip import stream, check
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
K_CONV, K_WIDE, K_DW, K_BNF, K_BNB, K_ADD, K_CAST, K_WSPLIT, K_NARROW = 1, 2, 3, 4, 5, 6, 7, 9, 10
F_BF16, F_TO_BF16, F_SIDE, F_JOIN = 1, 2, 4, 8
_ALIGN = 256
BF16 = torch.bfloat16

stats = {"passes": 0, "fallbacks": 0, "templates": 0}
# weight gradients on the library's second stream (nothing later in a backward list reads them): they fill the CUs
# the tails and small launches of the input-gradient chain leave idle
import os  # noqa: E402
dw_side_stream = os.environ.get("AABR_PLAN_DW_SIDE", "1") != "0"
# the lateral 1x1x1 convolutions of the FPN (they read a down-path map long before the up path needs them) can go
# there too (AABR_PLAN_JOIN in front of their reader); measured on the bench step: 14.65 -> 14.60 ms, inside the
# noise, so off by default
lateral_side_stream = os.environ.get("AABR_PLAN_LATERAL_SIDE", "0") != "0"
# a residual / lateral add whose second operand comes straight from a wide-kernel convolution rides in that
# convolution's write-out (forward only; the backward list is unchanged)
fuse_adds = os.environ.get("AABR_PLAN_FUSE_ADDS", "1") != "0"
# a training-mode BatchNorm right behind a wide-kernel convolution takes its statistics' partial sums from that
# convolution's write-out (aabr_conv_forward_wide_stats -> aabr_bn_forward_parts): one pass over the matrix less
# ... the same in bf16 storage (round 6; forward adds, and the gradient sums of the backward list)
fuse_adds_bf16 = os.environ.get("AABR_PLAN_FUSE_ADDS_BF16", "1") != "0"
conv_bn_stats = os.environ.get("AABR_PLAN_CONV_BN_STATS", "1") != "0"
# ... and a BatchNorm backward right behind the wide-kernel input-gradient launch that produced its d_out takes its
# statistics from that launch's write-out (aabr_conv_forward_wide[_bf16]_bwd_stats -> aabr_bn_backward_parts[_bf16])
conv_bn_bwd_stats = os.environ.get("AABR_PLAN_CONV_BN_BWD_STATS", "1") != "0"
# ... the same two from the write-out of the 32 -> 32 kernel (csrc/conv_narrow.hip; bf16 storage).  Measured on the config-4
# step, same box, two runs each: narrow kernel off 10.22 / 10.18 ms, on 10.02 / 10.01, on with these statistics 10.13 / 10.13 --
# the wave-level reductions (16 x 8 fp64 values per 16-row group) cost the launch more than the BatchNorm's own HBM-bound
# statistics passes; off by default.
narrow_stats = os.environ.get("AABR_PLAN_NARROW_STATS", "0") != "0"


# Data-parallel hook (extension; the reference wraps the model in DistributedDataParallel, whose bucketed all-reduce
# starts while backward is still running, tools/train_net_sparse3d.py:64-69): with `grad_segments` = S > 1 and
# `on_grads_ready` set, the compiled backward list is handed to the library in S pieces and after each piece the hook
# receives the gradients that are final by then -- `on_grads_ready(piece, S, flat, [(parameter, gradient view)])`,
# `flat` = the contiguous slice of the pass's gradient buffer that holds exactly those views -- so a collective on
# `flat` runs underneath the rest of the backward pass.  Same launches in the same order; a piece boundary only
# makes the caller's stream wait for the weight gradients issued so far on the library's second stream.
grad_segments = 0
on_grads_ready = None
# Pipelined hand-over (aabr_plan_submit / aabr_plan_drain; option): every `pipeline_records` records the part of the list
# that is final goes to the library's launcher thread, which issues its launches while this thread fills the next part --
# the two halves of a pass's host cost (filling records, issuing launches) overlap instead of adding up, and the device
# starts the pass one part in instead of after the whole list.  Same records in the same order, bit-equal results
# (tests/test_gpu_fpn.py).  Measured on the bench step (round 4, tools/tools_host_breakdown.py): the two passes' host time
# 2.50 -> 2.00 ms (the launcher needs 1.37 ms per step to issue ~450 launches either way, and the pass must be drained
# before torch's next launch), step time unchanged within noise in fp32 (device-bound) and bf16 (7.88 vs 7.90 ms: the
# device's chain of small launches is as long as the host's) -- so the default is 0 = one call per pass.
pipeline_records = int(os.environ.get("AABR_PLAN_PIPELINE", "0"))
# tests: a list here receives every _Pass that runs (its arena holds every activation of the pass --
# `_Pass.bn_outputs()`); None in production
debug_passes = None
# a pass that will not be differentiated packs its activations by liveness (AABR_PLAN_PACK_ARENA=0: one slot each, as
# the training pass needs them)
pack_inference_arena = os.environ.get("AABR_PLAN_PACK_ARENA", "1") != "0"


class Unsupported(Exception):
    pass


def _es(dt):
    return 2 if dt == BF16 else 4


def _dp(t):
    return t.data_ptr() if t is not None else 0


# buffer references inside a backward list: (space, index); spaces resolved to address lists per pass
F, G, E = 0, 1, 2      # forward arena (index 0 = the graph input) / gradient arena / external output gradients


class _Template(object):
    """the structure of the graph, independent of the scene"""

    def __init__(self, net, x_spatial, x_planes, dtype, plan):
        self.dtype = dtype
        self.levels, self.lvl = [], {}
        self.fbufs = []                      # (level, planes, dtype)
        self.books, self.bidx = [], {}       # rule books: (kind, dict key, builder args)
        self.fops = []
        self.params, self.pidx = [], {}
        self.stat_floats = 0
        self.bn_floats = 0
        self.max_planes = 1
        self.hidden = []                     # (level, planes) of every convolution output (hidden-state counter)
        self.macs = []                       # (book, weight) per convolution, in order
        self.bns = []
        self.branches = []
        self.packs = {id(w): (pf.data_ptr(), pt.data_ptr()) for w, (pf, pt) in zip(plan.weights, plan.packs)}
        self.lib = _hip.load()
        x = (self._new(self._level(x_spatial), x_planes, torch.float32), x_spatial)
        xs = self.cast(x, dtype)
        rpn, roi = _fpn_graph(net, self, xs)
        self.n_rpn = len(rpn)
        self.outs = [self.cast(t, torch.float32) for t in rpn] + [self.cast(t, torch.float32) for t in roi]
        self._bwd = {}
        # forward emission order: a lateral branch (ops that depend on one earlier buffer only) is issued right after
        # that buffer's producer, on the second stream; its first reader joins
        self.emit = self._emission()
        self.fuse = self._fusable_adds() if (fuse_adds and not lateral_side_stream) else {}
        stats["templates"] += 1

    def _fusable_adds(self):
        """{id(conv op): (add op, other operand)}: a convolution whose only reader is an add with an operand that
        exists before the convolution runs -- the add can ride in the convolution's write-out when the pass dispatches it
        to the wide kernel or the offset split (`aabr_conv_forward_wide_res`, `aabr_conv_forward_wide_bf16_res`, the
        split's second stage; a + b == b + a bit for bit, and in bf16 storage the write-out rounds the convolution's
        value before the sum exactly as the separate add would have read it)"""
        uses, prod = {}, {0: -1}
        for i, op in enumerate(self.fops):
            ins = (op[1], op[2]) if op[0] == "add" else (op[1],)
            for b in ins:
                uses[b] = uses.get(b, 0) + 1
            prod[_out_of(op)] = i
        for b, _ in self.outs:
            uses[b] = uses.get(b, 0) + 1
        out = {}
        for i, op in enumerate(self.fops):
            if op[0] != "add" or (op[6] and not fuse_adds_bf16):
                continue
            for conv_b, other in ((op[2], op[1]), (op[1], op[2])):
                j = prod.get(conv_b, -1)
                if j >= 0 and self.fops[j][0] == "conv" and uses.get(conv_b) == 1 and prod.get(other, 1 << 30) < j \
                        and id(self.fops[j]) not in out:
                    out[id(self.fops[j])] = (op, other)
                    break
        return out

    def branch(self, start, after_buf):
        """ops fops[start:] so far form a branch that reads only `after_buf`; the NEXT op traced is its reader"""
        self.branches.append((start, len(self.fops), after_buf))

    def _emission(self):
        order = [(i, 0) for i in range(len(self.fops))]
        if not (lateral_side_stream and self.branches):
            return [(self.fops[i], f) for i, f in order]
        moved, joins, after = set(), set(), {}
        for start, end, buf in self.branches:
            prod = [i for i, op in enumerate(self.fops) if _out_of(op) == buf]
            if len(prod) != 1 or prod[0] >= start or end >= len(self.fops):
                continue
            moved.update(range(start, end))
            joins.add(end)
            after.setdefault(prod[0], []).extend(range(start, end))
        out = []
        for i in range(len(self.fops)):
            if i in moved:
                continue
            out.append((self.fops[i], F_JOIN if i in joins else 0))
            for j in after.get(i, ()):
                out.append((self.fops[j], F_SIDE))
        return out

    # ---- bookkeeping -------------------------------------------------------------------------------------------
    def signature(self, plan):
        return (plan.arena.data_ptr(), tuple(p.data_ptr() for p in self.params),
                tuple(m.running_mean.data_ptr() for m in self.bns), tuple(m.running_var.data_ptr() for m in self.bns))

    def _level(self, sp):
        i = self.lvl.get(sp)
        if i is None:
            i = self.lvl[sp] = len(self.levels)
            self.levels.append(sp)
        return i

    def _new(self, lvl, planes, dt):
        self.fbufs.append((lvl, planes, dt))
        return len(self.fbufs) - 1

    def _param(self, p):
        """one gradient slot per parameter, STORED by its one dW / BatchNorm-backward launch: a parameter used by two
        operators of the graph (tied weights) would need a sum there, so such a network goes through the modules"""
        i = self.pidx.get(id(p))
        if i is not None:
            raise Unsupported("a parameter shared by two operators (tied weights)")
        i = self.pidx[id(p)] = len(self.params)
        self.params.append(p)
        return i

    def _book(self, kind, key, args):
        i = self.bidx.get((kind, key))
        if i is None:
            i = self.bidx[(kind, key)] = len(self.books)
            self.books.append((kind, key, args))
        return i

    # ---- symbolic forward: one handler per module type ----------------------------------------------------------
    def run(self, m, x):
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
        raise Unsupported(type(m).__name__)

    def add(self, xs):
        """left-to-right sum (utils._sum_features)"""
        acc = xs[0]
        for t in xs[1:]:
            lvl, planes, dt = self.fbufs[acc[0]]
            assert self.fbufs[t[0]] == (lvl, planes, dt) and t[1] == acc[1]
            y = self._new(lvl, planes, dt)
            self.fops.append(("add", acc[0], t[0], y, lvl, planes, F_BF16 if dt == BF16 else 0))
            acc = (y, acc[1])
        return acc

    def cast(self, x, dtype):
        lvl, planes, dt = self.fbufs[x[0]]
        if dt == dtype:
            return x
        y = self._new(lvl, planes, dtype)
        self.fops.append(("cast", x[0], y, lvl, planes, F_TO_BF16 if dtype == BF16 else 0))
        return (y, x[1])

    def bn(self, m, x):
        if not (m.training or m.track_running_stats):
            raise Unsupported("BatchNormalization on batch statistics in evaluation mode")
        lvl, planes, dt = self.fbufs[x[0]]
        assert planes == m.nPlanes, (planes, m.nPlanes)
        y = self._new(lvl, planes, dt)
        st = self.stat_floats
        self.stat_floats += 2 * planes
        w = m.weight if m.affine else None
        b = m.bias if m.affine else None
        if m.affine:
            self._param(w)
            self._param(b)
        self.bns.append(m)
        self.bn_floats = max(self.bn_floats, int(self.lib.aabr_bn_scratch_floats(planes)))
        self.max_planes = max(self.max_planes, planes)
        self.fops.append(("bn", x[0], y, lvl, planes, F_BF16 if dt == BF16 else 0, 1 if m.training else 0,
                          float(m.eps), float(m.momentum), float(m.leakiness), st, _dp(m.running_mean),
                          _dp(m.running_var), _dp(w), _dp(b), m))
        return (y, x[1])

    def _conv_common(self, m, x, out_spatial, book, side_fwd, side_din, din_flags, side_dw):
        if hasattr(m, "bias") or getattr(m, "groups", 1) != 1:
            raise Unsupported("convolution with bias / groups")
        lvl, planes, dt = self.fbufs[x[0]]
        assert planes == m.nIn, (planes, m.nIn)
        w = m.weight
        if id(w) not in self.packs or not w.is_contiguous():
            raise Unsupported("weight packs missing (FPN_Net.prepack_weights)")
        lo = self._level(out_spatial)
        y = self._new(lo, m.nOut, dt)
        self._param(w)
        self.macs.append((book, w))
        self.hidden.append((lo, m.nOut))
        pf, pt = self.packs[id(w)]
        self.fops.append(("conv", x[0], y, lvl, lo, m.nIn, m.nOut, book, side_fwd, w.data_ptr(), pf,
                          side_din, din_flags, side_dw, pt, m))
        return (y, out_spatial)

    def subm(self, m, x):
        fs = SCN._key(m.filter_size)
        book = self._book("s", x[1] + fs, (x[1], m.filter_size))
        return self._conv_common(m, x, x[1], book, 0, 0, 1 | 2, 0)

    def conv(self, m, x):
        fs, st = SCN._key(m.filter_size), SCN._key(m.filter_stride)
        osz = tuple((i - f) // s + 1 for i, f, s in zip(x[1], fs, st))
        assert all((o - 1) * s + f == i for o, s, f, i in zip(osz, st, fs, x[1])), (x[1], osz, fs, st)
        book = self._book("c", x[1] + fs + st, (x[1], osz, m.filter_size, m.filter_stride))
        return self._conv_common(m, x, osz, book, 0, 1, 1, 0)

    def deconv(self, m, x):
        fs, st = SCN._key(m.filter_size), SCN._key(m.filter_stride)
        osz = tuple((i - 1) * s + f for i, f, s in zip(x[1], fs, st))
        book = self._book("c", osz + fs + st, (osz, x[1], m.filter_size, m.filter_stride))
        return self._conv_common(m, x, osz, book, 1, 0, 1, 1)

    # ---- the backward list for one pattern of incoming gradients -------------------------------------------------
    def backward(self, pattern, need_dx):
        key = (pattern, need_dx, tuple(p.requires_grad for p in self.params))
        b = self._bwd.get(key)
        if b is None:
            b = self._bwd[key] = self._make_backward(pattern, need_dx)
        return b

    def _make_backward(self, pattern, need_dx):
        gbufs, ops = [], []
        poff, ptotal = {}, 0

        def gnew(lvl, planes, dt):
            gbufs.append((lvl, planes, dt))
            return (G, len(gbufs) - 1)

        def pslot(p):
            nonlocal ptotal
            i = self.pidx[id(p)]
            if i not in poff:
                poff[i] = ptotal
                ptotal += (p.numel() + 63) // 64 * 64
            return poff[i] * 4

        acc = {}           # forward buffer -> the gradient accumulated so far (contributions in arrival order:
                           # autograd's order, the consumer created last delivers first; ((c0 + c1) + c2) ...)

        def arrive(b, ref):
            """an existing gradient buffer contributes to forward buffer b"""
            cur = acc.get(b)
            if cur is None:
                acc[b] = ref
                return
            lvl, planes, dt = self.fbufs[b]
            s = gnew(lvl, planes, dt)
            ops.append(("add", cur, ref, s, lvl, planes, F_BF16 if dt == BF16 else 0))
            acc[b] = s

        for k, ((b, _), has) in enumerate(zip(self.outs, pattern)):
            if has:
                arrive(b, (E, k))
        total = acc.get
        fuse = fuse_adds

        for op in reversed(self.fops):
            kind = op[0]
            if kind == "add":
                _, a_, b_, y, lvl, planes, flg = op
                gy = total(y)
                if gy is not None:       # same order as autograd's AddBackward: first operand, then second
                    arrive(a_, gy)
                    arrive(b_, gy)
            elif kind == "cast":
                _, x, y, lvl, planes, flg = op
                gy = total(y)
                if gy is None:
                    continue
                dtx = self.fbufs[x][2]
                gx = gnew(lvl, planes, dtx)
                ops.append(("cast", gy, gx, lvl, planes, F_TO_BF16 if dtx == BF16 else 0))
                arrive(x, gx)
            elif kind == "bn":
                _, x, y, lvl, planes, flg, train, eps, mom, leak, st, p_rm, p_rv, p_w, p_b, m = op
                gy = total(y)
                if gy is None:
                    continue
                gx = gnew(lvl, planes, self.fbufs[x][2])
                pw = pslot(m.weight) if m.affine and m.weight.requires_grad else -1
                pb = pslot(m.bias) if m.affine and m.bias.requires_grad else -1
                res = acc.get(x) if (fuse and (not flg or fuse_adds_bf16)) else None
                ops.append(("bn", x, gx, y, gy, lvl, planes, flg, leak, st, p_w, pw, pb, p_b, res))
                if res is not None:      # the sum with what has arrived so far rides in the apply pass
                    acc[x] = gx
                else:
                    arrive(x, gx)
            else:
                _, x, y, lvl, lo, n_in, n_out, book, side_fwd, p_w, pf, side_din, din_flags, side_dw, pt, m = op
                gy = total(y)
                if gy is None:
                    continue
                dt = self.fbufs[x][2]
                flg = F_BF16 if dt == BF16 else 0
                if x != 0 or need_dx:
                    gx = gnew(lvl, n_in, dt)
                    res = acc.get(x) if (fuse and (not flg or fuse_adds_bf16)) else None
                    tmp = gnew(lvl, n_in, dt) if res is not None else None   # used when the launch is not a wide one
                    # the launch reads d_out (n_out planes) and writes d_in (n_in planes)
                    ops.append(("din", gy, gx, lo, lvl, n_out, n_in, book, side_din, din_flags, p_w, pt, flg, res,
                                tmp))
                    if res is not None:
                        acc[x] = gx
                    else:
                        arrive(x, gx)
                if m.weight.requires_grad:
                    ops.append(("dw", x, gy, lo, n_in, n_out, book, side_dw, pslot(m.weight), flg))
        gx0 = total(0) if need_dx else None
        return {"gbufs": gbufs, "ops": ops, "poff": poff, "ptotal": ptotal, "gx0": gx0}


def _out_of(op):
    return op[3] if op[0] == "add" else op[2]


def _offsets(bufs, V, first=0):
    """arena offsets of [rows(level), planes] matrices; bufs[:first] are not in the arena"""
    offs, total = [0] * len(bufs), 0
    for i in range(first, len(bufs)):
        lvl, planes, dt = bufs[i]
        offs[i] = total
        total += (V[lvl] * planes * (2 if dt == BF16 else 4) + _ALIGN - 1) // _ALIGN * _ALIGN
    return offs, total


class _Pass(object):
    """one pass through the graph: the template bound to a scene"""

    def __init__(self, tpl, md, x):
        self.t, self.md, self.x = tpl, md, x
        self.dev = x.device
        self.lib = tpl.lib
        self.V = V = [md.grids[sp].V for sp in tpl.levels]
        assert V[tpl.fbufs[0][0]] == x.size(0)
        # rule books of the pass (built by FPN_Net._prebuild_geometry; looked up by their cache keys)
        bk = []
        for kind, key, args in tpl.books:
            tb = (md.submanifold if kind == "s" else md.rulebooks).get(key)
            if tb is None:
                tb = md.getSubmanifoldRuleBook(*args) if kind == "s" else md.getRuleBook(*args)
            bk.append((tb.out, tb.inn if tb.inn is not None else tb.out, tb))
        self.books = bk
        self._wide = {}
        self._tmp = []

    def wide_rows(self, n_in, n_out, rows_in, rows_out, vol, bf=False):
        key = (n_in, n_out, rows_in, rows_out, vol, bf)
        T = self._wide.get(key)
        if T is None:
            T = self._wide[key] = SCN.wide_tile_rows(n_in, n_out, rows_in, rows_out, vol, bf)
        return T

    def split_of(self, n_in, n_out, rows_in, rows_out, vol, bf=False):
        key = ("split", n_in, n_out, rows_in, rows_out, vol, bf)
        v = self._wide.get(key, 0)
        if v == 0:
            v = self._wide[key] = SCN.wide_split(n_in, n_out, rows_in, rows_out, vol, bf) or ()
        return v or None

    def res_ok(self, n_in, n_out, rows_in, rows_out, vol, bf=False):
        """a launch that can add a residual in its write-out: the wide kernel, or the offset split (its second stage,
        k_split_reduce, adds it)"""
        return bool(self.wide_rows(n_in, n_out, rows_in, rows_out, vol, bf) or
                    self.split_of(n_in, n_out, rows_in, rows_out, vol, bf))

    def conv_launch(self, pack, buf, off, src, rows_in, n_in, dst, rows_out, n_out, gather, p_w, p_pack, flags, bf,
                    xf=0, res=0):
        """the record of the launch SCN._conv_fwd makes for a prepacked weight; returns the new write offset"""
        self._lw = 0      # tile rows when the record is a K_WIDE one (its write-out can form BatchNorm statistics)
        if rows_out == 0:
            return off
        if not res and SCN.narrow_ok(n_in, n_out, rows_in, rows_out, gather.vol, bf):
            # 32 -> 32 planes from the gather table, raw weights (csrc/conv_narrow.hip): the choice SCN._conv_fwd makes first
            pack(buf, off, K_NARROW, xf | (F_BF16 if bf else 0), n_in, n_out, gather.vol, flags & 3, 0, 0, 0.0, 0.0, 0.0,
                 0.0, rows_in, rows_out, 0, 0, src, dst, gather.table.data_ptr(), p_w, 0, 0, 0, 0, 0, 0, 0, 0)
            if bf and narrow_stats:   # its write-out can form BatchNorm statistics too: one part per workgroup (negative = a part COUNT)
                self._lw = -int(self.lib.aabr_conv_narrow_parts(rows_out))
            return off + 176
        T = self.wide_rows(n_in, n_out, rows_in, rows_out, gather.vol, bf)
        self._lw = T
        sp = None if T else self.split_of(n_in, n_out, rows_in, rows_out, gather.vol, bf)
        assert T or sp or not res
        if T:
            pack(buf, off, K_WIDE, xf | (F_BF16 if bf else 0), n_in, n_out, gather.vol, flags & 3, T, 0, 0.0, 0.0, 0.0,
                 0.0, rows_in, rows_out, 0, 0, src, dst, gather.blocks_wide(T).data_ptr(), res, 0, p_pack, 0, 0, 0, 0,
                 0, 0)
        elif sp:       # coarse map: the wide kernel cut into parts over the filter offsets (fp32 storage)
            Ts, P = sp
            # its own scratch per record: the list is launched later in ONE call, so a shared grow-only workspace could
            # be reallocated under records already written (the allocator frees it in stream order once the pass's
            # next list is built)
            tmp = torch.empty(P * rows_out * n_out, dtype=torch.float32, device=self.dev)
            self._tmp.append(tmp)
            ws = tmp.data_ptr()
            pack(buf, off, K_WSPLIT, xf | (F_BF16 if bf else 0), n_in, n_out, gather.vol, flags & 3, Ts, P, 0.0, 0.0, 0.0, 0.0, rows_in, rows_out,
                 0, 0, src, dst, gather.blocks_wide(Ts).data_ptr(), res, 0, p_pack, ws, 0, 0, 0, 0, 0)
        else:
            pack(buf, off, K_CONV, (F_BF16 if bf else 0) | xf, n_in, n_out, gather.vol, flags | 4, 0, 0, 0.0, 0.0, 0.0, 0.0,
                 rows_in, rows_out, 0, 0, src, dst, gather.blocks().data_ptr(), p_w, 0, p_pack, 0, 0, 0, 0, 0, 0)
        return off + 176

    def _live_offsets(self):
        """arena offsets for a pass nobody will differentiate (torch.no_grad): a buffer's bytes are handed on once its
        last reader has been recorded -- first fit over a free list, in record order (one stream) -- instead of every
        activation of the network living until the pass object dies.  Returns (offsets, total, total without reuse)."""
        t, V = self.t, self.V
        fbufs, books, fuse = t.fbufs, self.books, t.fuse
        steps, skip = [], set()            # (buffers read, buffer written) per emitted record
        for op, xf in t.emit:
            if xf & F_SIDE:
                return None
            kind = op[0]
            if kind == "conv":
                x, y, lvl, lo, n_in, n_out, book, side = op[1:9]
                fz = fuse.get(id(op))
                if fz is not None and V[lo] and self.res_ok(n_in, n_out, V[lvl], V[lo], books[book][side].vol,
                                                            fbufs[x][2] == BF16):
                    add_op, other = fz
                    skip.add(id(add_op))
                    steps.append(((x, other), add_op[3]))
                else:
                    steps.append(((x,), y))
            elif kind == "add":
                if id(op) not in skip:
                    steps.append(((op[1], op[2]), op[3]))
            else:
                steps.append(((op[1],), op[2]))
        last = {}
        for i, (rd, _) in enumerate(steps):
            for b in rd:
                last[b] = i
        for b, _ in t.outs:
            last[b] = len(steps)
        size = lambda b: (V[fbufs[b][0]] * fbufs[b][1] * _es(fbufs[b][2]) + _ALIGN - 1) // _ALIGN * _ALIGN
        offs, free, top = [0] * len(fbufs), [], 0      # free: [offset, bytes], sorted by offset
        dying = {}
        for b, i in last.items():
            dying.setdefault(i, []).append(b)
        for i, (rd, wr) in enumerate(steps):
            need = size(wr)
            for f in free:                               # first fit
                if f[1] >= need:
                    offs[wr] = f[0]
                    f[0] += need
                    f[1] -= need
                    break
            else:
                offs[wr] = top
                top += need
            free = [f for f in free if f[1] > 0]
            if wr not in last:                           # written, never read, not an output: free at once
                dying.setdefault(i, []).append(wr)
            for b in dying.get(i, ()):                   # read for the last time by this record
                if b == 0:
                    continue                             # the graph input is not in the arena
                free.append([offs[b], size(b)])
            free.sort()
            merged = []
            for f in free:                               # coalesce neighbours
                if merged and merged[-1][0] + merged[-1][1] == f[0]:
                    merged[-1][1] += f[1]
                else:
                    merged.append(f)
            free = merged
        return offs, top, sum(size(b) for b in range(1, len(fbufs)))

    def forward(self, keep=True):
        """`keep` False (no autograd): the arena is packed by liveness (`_live_offsets`)"""
        try:
            return self._forward(keep)
        except BaseException:
            self.lib.aabr_plan_drain()     # never leave parts of a failed pass in the launcher's queue
            raise

    def _forward(self, keep):
        t, V = self.t, self.V
        self._tmp = []
        packed = None if keep else self._live_offsets()
        if packed is not None:
            offs, total, flat = packed
            stats["arena_bytes_packed"], stats["arena_bytes_flat"] = total, flat
        else:
            offs, total = _offsets(t.fbufs, V, 1)
        self.arena = torch.empty(max(total, 1), dtype=torch.uint8, device=self.dev)
        self.stat = torch.empty(max(t.stat_floats, 1), dtype=torch.float32, device=self.dev)
        base, sbase = self.arena.data_ptr(), self.stat.data_ptr()
        self.offs = offs
        A = self.A = [base + o for o in offs]
        A[0] = self.x.data_ptr()
        bnws = _hip.workspace("bn", t.bn_floats, torch.float32, self.dev).data_ptr() if t.bn_floats else 0
        buf = bytearray(len(t.fops) * 176)
        pack, off = _OP.pack_into, 0
        books = self.books
        fbufs = t.fbufs
        fuse, skip = t.fuse, set()
        # BatchNorm statistics from the producing convolution's write-out: a training-mode BatchNorm that is the NEXT
        # record on its stream after the k_conv_cs launch that wrote its input gets the per-tile partial sums from
        # that launch (record p6 -> BatchNorm p9) and skips its own statistics pass over the matrix
        last = {0: None, F_SIDE: None}     # per stream: (buffer index written, record offset, tile rows, planes)
        cstat = {}
        part, start, strm = pipeline_records * 176, 0, stream()
        for op, xf in t.emit:
            if part and off - start >= part:
                # records up to `safe` are final (a convolution record stays open while the BatchNorm behind it may
                # still claim its write-out for the statistics)
                safe = off
                for lw in last.values():
                    if lw is not None and lw[1] < safe:
                        safe = lw[1]
                if safe > start:
                    check(self.lib.aabr_plan_submit(bytes(buf[start:safe]), (safe - start) // 176, strm, 1))
                    start = safe
            kind = op[0]
            sk = xf & F_SIDE
            if kind == "conv":
                x, y, lvl, lo, n_in, n_out, book, side, p_w, pf = op[1:11]
                fz = fuse.get(id(op))
                bfx = fbufs[x][2] == BF16
                if fz is not None and V[lo] and self.res_ok(n_in, n_out, V[lvl], V[lo], books[book][side].vol, bfx):
                    add_op, other = fz           # out = conv + other, written where the add would have written
                    skip.add(id(add_op))
                    off0 = off
                    off = self.conv_launch(pack, buf, off, A[x], V[lvl], n_in, A[add_op[3]], V[lo], n_out,
                                           books[book][side], p_w, pf, 0, bfx, xf, A[other])
                    last[sk] = (add_op[3], off0, self._lw, n_out) if (self._lw >= 64 or self._lw < 0) else None
                else:
                    off0 = off
                    off = self.conv_launch(pack, buf, off, A[x], V[lvl], n_in, A[y], V[lo], n_out, books[book][side],
                                           p_w, pf, 0, fbufs[x][2] == BF16, xf)
                    last[sk] = (y, off0, self._lw, n_out) if (self._lw >= 64 or self._lw < 0) else None
                continue
            elif kind == "bn":
                _, x, y, lvl, planes, flg, train, eps, mom, leak, st, p_rm, p_rv, p_w, p_b, m = op
                if V[lvl]:
                    parts, nparts, lw = 0, 0, last[sk]
                    if conv_bn_stats and train and lw is not None and lw[0] == x and lw[3] == planes:
                        nparts = -lw[2] if lw[2] < 0 else (V[lvl] + lw[2] - 1) // lw[2]
                        ws = cstat.get(sk)
                        if ws is None:   # one buffer per stream: written by the convolution, read by the very next record
                            ws = cstat[sk] = _hip.workspace("conv_stats%d" % sk, (max(V) // 64 + 1) * 2 * t.max_planes,
                                                            torch.float64, self.dev).data_ptr()
                        parts = ws
                        struct.pack_into("<Q", buf, lw[1] + 128, ws)          # the convolution record's p6
                    pack(buf, off, K_BNF, flg | xf, planes, train, nparts, 0, 0, 0, eps, mom, leak, 0.0, V[lvl], 0, 0, 0,
                         A[x], A[y], sbase + st * 4, sbase + (st + planes) * 4, p_rm, p_rv, p_w, p_b, bnws, parts, 0, 0)
                    off += 176
            elif kind == "add":
                if id(op) in skip:
                    continue
                _, a_, b_, y, lvl, planes, flg = op
                pack(buf, off, K_ADD, flg | xf, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lvl] * planes, 0, 0, 0,
                     A[a_], A[b_], A[y], 0, 0, 0, 0, 0, 0, 0, 0, 0)
                off += 176
            else:
                _, x, y, lvl, planes, flg = op
                pack(buf, off, K_CAST, flg | xf, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lvl] * planes, 0, 0, 0,
                     A[x], A[y], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
                off += 176
            last[sk] = None
        if part:
            check(self.lib.aabr_plan_submit(bytes(buf[start:off]), (off - start) // 176, strm, 0))
            check(self.lib.aabr_plan_drain())      # everything issued: the caller's next launches queue behind the pass
        elif off:
            check(self.lib.aabr_plan_run(bytes(buf[:off]), off // 176, strm))
        res = []
        for b, _ in t.outs:
            lvl, planes, dt = fbufs[b]
            n = V[lvl] * planes * _es(dt)
            res.append(self.arena[offs[b]:offs[b] + n].view(dt).view(V[lvl], planes))
        return res

    def buffer(self, b):
        """forward buffer `b` of this pass as a tensor view into the arena (tests)"""
        lvl, planes, dt = self.t.fbufs[b]
        n = self.V[lvl] * planes * _es(dt)
        return self.arena[self.offs[b]:self.offs[b] + n].view(dt).view(self.V[lvl], planes)

    def bn_outputs(self):
        """{BatchNorm module: its output matrix of this pass} (tests: the compiled graph runs no module hooks)"""
        return {op[-1]: self.buffer(op[2]) for op in self.t.fops if op[0] == "bn" and self.V[op[3]]}

    @staticmethod
    def _pinv(bw, byte_off):
        """parameter index of a gradient slot (byte offset into the pass's gradient buffer)"""
        inv = bw.get("pinv")
        if inv is None:
            inv = bw["pinv"] = {o * 4: i for i, o in bw["poff"].items()}
        return inv[byte_off]

    def backward(self, gouts, need_dx):
        try:
            return self._backward(gouts, need_dx)
        except BaseException:
            self.lib.aabr_plan_drain()
            raise

    def _backward(self, gouts, need_dx):
        t, V, A = self.t, self.V, self.A
        self._tmp = []
        gouts = [g.contiguous() if g is not None else None for g in gouts]
        bw = t.backward(tuple(g is not None for g in gouts), need_dx)
        gbufs, bops = bw["gbufs"], bw["ops"]
        offs, total = _offsets(gbufs, V)
        garena = torch.empty(max(total, 1), dtype=torch.uint8, device=self.dev)
        # the data-parallel hook all-reduces slices of this buffer, 64-float padding between the slots included
        hooked = grad_segments > 1 and on_grads_ready is not None
        gparams = (torch.zeros if hooked else torch.empty)(max(bw["ptotal"], 1), dtype=torch.float32, device=self.dev)
        gbase, pbase, sbase = garena.data_ptr(), gparams.data_ptr(), self.stat.data_ptr()
        AD = (A, [gbase + o for o in offs], [_dp(g) for g in gouts])
        for (b, _), g in zip(t.outs, gouts):
            if g is not None:
                lvl, planes, dt = t.fbufs[b]
                assert g.dtype == dt and g.numel() == V[lvl] * planes
        books = self.books
        # weight-gradient scratch: the largest of the pass
        dws, dwmc = 0, []
        for op in bops:
            if op[0] == "dw":
                mc = books[op[6]][op[7]].max_chunks(op[4], op[5])
                dwmc.append(mc)
                dws = max(dws, mc * op[4] * op[5])
        dwmc.reverse()
        dwws = _hip.workspace("dw", dws, torch.float32, self.dev).data_ptr() if dws else 0
        bnws = _hip.workspace("bn", t.bn_floats, torch.float32, self.dev).data_ptr() if t.bn_floats else 0
        buf = bytearray(len(bops) * 2 * 176)
        pack, off = _OP.pack_into, 0
        dw_side = F_SIDE if dw_side_stream else 0
        # pieces of the list for the data-parallel hook: cut after every len/S-th record; parameter gradients are laid
        # out in the order their records appear (pslot), so the ones a piece completes form one slice of `gparams`
        nseg = grad_segments if (grad_segments > 1 and on_grads_ready is not None) else 1
        cut = [(len(bops) * (q + 1)) // nseg for q in range(nseg)] if nseg > 1 else []
        inv = sorted((o, i) for i, o in bw["poff"].items()) if nseg > 1 else []
        done_floats, seg_no, next_inv = 0, 0, 0
        last_din, bstat = None, None     # the last wide input-gradient record of the main stream / statistics workspace
        part, start, strm = (pipeline_records * 176 if nseg == 1 else 0), 0, stream()
        for op_no, op in enumerate(bops):
            if part and off - start >= part:
                safe = last_din[1] if last_din is not None else off    # (that record may still get the statistics)
                if safe > start:
                    check(self.lib.aabr_plan_submit(bytes(buf[start:safe]), (safe - start) // 176, strm, 1))
                    start = safe
            if nseg > 1 and seg_no < nseg - 1 and op_no == cut[seg_no]:
                last_din = None          # its record has been handed over
                if off:
                    check(self.lib.aabr_plan_run(bytes(buf[:off]), off // 176, stream()))
                    off = 0
                pairs = []
                while next_inv < len(inv) and inv[next_inv][0] < done_floats:
                    o, i = inv[next_inv]
                    pairs.append((t.params[i], gparams[o:o + t.params[i].numel()].view_as(t.params[i])))
                    next_inv += 1
                if pairs:
                    lo = pairs[0][1].data_ptr() - pbase
                    on_grads_ready(seg_no, nseg, gparams[lo // 4:done_floats], pairs)
                seg_no += 1
            kind = op[0]
            if kind == "dw":
                done_floats = max(done_floats, op[8] // 4 + (t.params[self._pinv(bw, op[8])].numel() + 63) // 64 * 64)
            elif kind == "bn":
                for po in (op[11], op[12]):
                    if po >= 0:
                        done_floats = max(done_floats, po // 4 + (t.params[self._pinv(bw, po)].numel() + 63) // 64 * 64)
            if kind == "din":
                _, gy, gx, lo, lvl, n_in, n_out, book, side, flags, p_w, pt, flg, res, tmp = op
                g = books[book][side]
                last_din = None
                if res is None:
                    off0 = off
                    off = self.conv_launch(pack, buf, off, AD[gy[0]][gy[1]], V[lo], n_in, AD[gx[0]][gx[1]], V[lvl],
                                           n_out, g, p_w, pt, flags, flg == F_BF16)
                    if (self._lw >= 64 or self._lw < 0) and off > off0:
                        last_din = (gx, off0, self._lw, n_out)
                elif V[lvl] and self.res_ok(n_in, n_out, V[lo], V[lvl], g.vol, flg == F_BF16):
                    off0 = off
                    off = self.conv_launch(pack, buf, off, AD[gy[0]][gy[1]], V[lo], n_in, AD[gx[0]][gx[1]], V[lvl],
                                           n_out, g, p_w, pt, flags, flg == F_BF16, 0, AD[res[0]][res[1]])
                    if (self._lw >= 64 or self._lw < 0) and off > off0:
                        last_din = (gx, off0, self._lw, n_out)
                else:                    # not a wide launch: d_in into the spare buffer, then the sum
                    off = self.conv_launch(pack, buf, off, AD[gy[0]][gy[1]], V[lo], n_in, AD[tmp[0]][tmp[1]], V[lvl],
                                           n_out, g, p_w, pt, flags, flg == F_BF16)
                    pack(buf, off, K_ADD, flg, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lvl] * n_out, 0, 0, 0,
                         AD[res[0]][res[1]], AD[tmp[0]][tmp[1]], AD[gx[0]][gx[1]], 0, 0, 0, 0, 0, 0, 0, 0, 0)
                    off += 176
            elif kind == "dw":
                _, x, gy, lo, n_in, n_out, book, side, pg, flg = op
                g = books[book][side]
                pack(buf, off, K_DW, flg | dw_side, n_in, n_out, g.vol, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lo], dwmc.pop(), 0, 0,
                     A[x], AD[gy[0]][gy[1]], g.pairs().data_ptr(), pbase + pg, 0, dwws, 0, 0, 0, 0, 0, 0)
                off += 176
            elif kind == "bn":
                _, x, gx, y, gy, lvl, planes, flg, leak, st, p_w, pw, pb, p_b, res = op
                if V[lvl]:
                    parts, nparts, ld = 0, 0, last_din
                    if (conv_bn_bwd_stats and ld is not None and ld[0] == gy and ld[3] == planes
                            and (flg == F_BF16 or (not flg and leak >= 0.0))):
                        # the BatchNorm's d_out was written by the wide-kernel input-gradient launch right before it (the
                        # weight-gradient record in between runs on the second stream): that launch's write-out forms the
                        # backward statistics (record i32[5] = 1, p6 stats, p7.. the BatchNorm's input and coefficients)
                        nparts = -ld[2] if ld[2] < 0 else (V[lvl] + ld[2] - 1) // ld[2]
                        if bstat is None:
                            bstat = _hip.workspace("conv_bwd_stats", (max(V) // 64 + 1) * 2 * t.max_planes,
                                                   torch.float64, self.dev).data_ptr()
                        parts = bstat
                        struct.pack_into("<i", buf, ld[1] + 8 + 5 * 4, 1)                                   # i32[5]
                        struct.pack_into("<f", buf, ld[1] + 32, leak)                                       # f32[0]
                        struct.pack_into("<6Q", buf, ld[1] + 80 + 6 * 8, bstat, A[x], sbase + st * 4,
                                         A[y] if flg == F_BF16 else sbase + (st + planes) * 4, p_w, p_b)    # p6 .. p11
                    pack(buf, off, K_BNB, flg, planes, nparts, 0, 0, 0, 0, 0.0, 0.0, leak, 0.0, V[lvl], parts, 0, 0,
                         A[x], AD[gx[0]][gx[1]], A[y], AD[gy[0]][gy[1]], sbase + st * 4, sbase + (st + planes) * 4,
                         p_w, pbase + pw if pw >= 0 else 0, pbase + pb if pb >= 0 else 0, bnws, p_b,
                         AD[res[0]][res[1]] if res is not None else 0)
                    off += 176
                last_din = None
            elif kind == "add":
                _, a_, b_, s, lvl, planes, flg = op
                pack(buf, off, K_ADD, flg, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lvl] * planes, 0, 0, 0,
                     AD[a_[0]][a_[1]], AD[b_[0]][b_[1]], AD[s[0]][s[1]], 0, 0, 0, 0, 0, 0, 0, 0, 0)
                off += 176
                last_din = None
            else:
                last_din = None
                _, gy, gx, lvl, planes, flg = op
                pack(buf, off, K_CAST, flg, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, V[lvl] * planes, 0, 0, 0,
                     AD[gy[0]][gy[1]], AD[gx[0]][gx[1]], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
                off += 176
        if part:
            check(self.lib.aabr_plan_submit(bytes(buf[start:off]), (off - start) // 176, strm, 0))
            check(self.lib.aabr_plan_drain())
        elif off:
            check(self.lib.aabr_plan_run(bytes(buf[:off]), off // 176, stream()))
        if nseg > 1:                      # the last piece's gradients
            pairs = []
            while next_inv < len(inv):
                o, i = inv[next_inv]
                pairs.append((t.params[i], gparams[o:o + t.params[i].numel()].view_as(t.params[i])))
                next_inv += 1
            if pairs:
                lo = pairs[0][1].data_ptr() - pbase
                on_grads_ready(nseg - 1, nseg, gparams[lo // 4:bw["ptotal"]], pairs)
        pgrad = [None] * len(t.params)
        for i, o in bw["poff"].items():
            p = t.params[i]
            pgrad[i] = gparams[o:o + p.numel()].view_as(p)
        dx = None
        gx0 = bw["gx0"]
        if gx0 is not None:
            lvl, planes, dt = t.fbufs[0]
            if gx0[0] == G:
                o = offs[gx0[1]]
                dx = garena[o:o + V[lvl] * planes * _es(dt)].view(dt).view(V[lvl], planes)
            else:   # the graph input reached an output untouched: its gradient is the incoming one
                dx = gouts[gx0[1]]
        return dx, pgrad


class _GraphFunction(Function):
    @staticmethod
    def forward(ctx, ps, x, *params):
        res = ps.forward()
        ctx.ps = ps
        ctx.need_dx = x.requires_grad
        # The backward list reads the transposed weight packs of THIS forward and recomputes the BatchNorm activation
        # signs from the BatchNorm parameters: state held by reference.  An in-place parameter update between forward
        # and backward (torch would raise "modified by an inplace operation") must not pass silently.
        ctx.versions = [p._version for p in params]
        ctx.params = params
        return tuple(res)

    @staticmethod
    def backward(ctx, *gouts):
        ps, ctx.ps = ctx.ps, None
        for p, v in zip(ctx.params, ctx.versions):
            if p._version != v:
                raise RuntimeError("a parameter of the compiled FPN_Net graph was modified in place between forward and "
                                   "backward (version %d -> %d): its packed copy / the BatchNorm coefficients the backward "
                                   "list reads belong to the forward pass" % (v, p._version))
        ctx.params = None
        dx, pgrad = ps.backward(gouts, ctx.need_dx)
        return (None, dx) + tuple(pgrad)


def _fpn_graph(net, ps, x):
    """forward_fpn (fpn_net.py:168-203), symbolically, in the module path's order"""
    n = len(net.m_downs)
    downs = []
    for m in net.m_downs:
        x = ps.run(m, x)
        downs.append(x)
    x = ps.run(net.m_shortcuts[-1], x)
    ups = [x]
    for k in range(net._top_down_levels()):      # all n - 1 stages (the reference), or up to the last consumed map
        j = n - 1 - k - 1
        x = ps.run(net.m_ups[k], x)
        mark = len(ps.fops)
        sc = ps.run(net.m_shortcuts[j], downs[j])
        ps.branch(mark, downs[j][0])
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


def _template(net, x_spatial, x_planes, plan):
    """the cached template of this network state (rebuilt when a mode flag, a parameter address, the weight-pack
    arena or the set of hooks changes)"""
    mods = net.__dict__.get("_graph_modules")
    if mods is None:
        mods = net.__dict__["_graph_modules"] = list(net.modules())
    mode = 0
    for m in mods:                 # any hook anywhere in the graph: the modules themselves must run
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            raise Unsupported("hooks")
        mode = mode * 2 + (1 if m.training else 0)
    cache = net.__dict__.setdefault("_graph_templates", {})
    key = (x_spatial, x_planes, net.feature_dtype, mode, bool(net.prune_unused_levels))
    tpl = cache.get(key)
    if tpl is not None and tpl.sig == tpl.signature(plan):
        return tpl
    tpl = _Template(net, x_spatial, x_planes, net.feature_dtype, plan)
    tpl.sig = tpl.signature(plan)
    if len(cache) > 8:
        cache.clear()
    cache[key] = tpl
    return tpl


def run_fpn(net, net1):
    """FPN_Net.forward after layers_in: (rpn_maps, roi_maps) as SparseConvNetTensors, or None when the graph
    holds something the executor does not handle (the caller then runs the modules)."""
    import sparseconvnet
    x = net1.features
    plan = net.__dict__.get("_pack_plan")
    if x.dim() != 2 or x.size(0) == 0 or x.dtype != torch.float32 or not x.is_contiguous() or plan is None or \
            plan.dtype != net.feature_dtype or plan._ptrs is None:
        return None
    try:
        tpl = _template(net, SCN._key(net1.spatial_size), x.size(1), plan)
    except Unsupported:
        stats["fallbacks"] += 1
        return None
    train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in tpl.params))
    ps = _Pass(tpl, net1.metadata, x)
    if debug_passes is not None:
        debug_passes.append(ps)
    if SCN.count_macs:
        for book, w in tpl.macs:
            sparseconvnet.forward_pass_multiplyAdd_count += SCN._macs(ps.books[book][2], w)
    sparseconvnet.forward_pass_hidden_states += sum(ps.V[lvl] * planes for lvl, planes in tpl.hidden)
    if train:
        feats = _GraphFunction.apply(ps, x, *tpl.params)
    else:
        feats = ps.forward(keep=not pack_inference_arena)
    stats["passes"] += 1
    res = []
    for (b, sp), f in zip(tpl.outs, feats):
        t = SparseConvNetTensor()
        t.metadata = net1.metadata
        t.spatial_size = torch.LongTensor(list(sp))
        t.features = f
        res.append(t)
    return res[:tpl.n_rpn], res[tpl.n_rpn:]

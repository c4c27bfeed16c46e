"""`sparseconvnet.SCN` -- the reference's native-extension namespace, re-implemented on the
MI355X C ABI (include/aabr_hip.h).

Function names, argument order and ownership rules are those of the reference's pybind11
module (SparseConvNet/sparseconvnet/SCN/pybind.cpp:11-235; dispatch in
SCN/sparseconvnet_cuda.cpp:281-455): the caller passes an EMPTY output tensor which is
resized and filled here; optional tensors arrive as empty tensors; d_weight / d_bias arrive
pre-zeroed.  What differs by design: all grid state (hash tables, site lists, rule tables)
lives in HBM inside `Metadata_3`, nothing is rebuilt or copied per layer, and every kernel
runs on the current PyTorch HIP stream.
"""
import torch

import _hip
from _hip import ptr, stream, check

count_macs = True  # the reference returns the multiply-add count of every conv (costs one
                   # small D2H read per rule book, cached afterwards)


def n_rulebook_bits():
    """pybind.cpp:234"""
    return 32


def _key(t):
    return tuple(int(x) for x in (t.tolist() if hasattr(t, "tolist") else t))


class _Grid(object):
    """one scale of the scene: site list + hash table (replaces SparseGrids, Metadata.h:24-34)"""
    __slots__ = ("coords", "keys", "vals", "cap", "V", "batch_size")

    def __init__(self, coords, keys, vals, cap, V):
        self.coords, self.keys, self.vals, self.cap, self.V = coords, keys, vals, cap, V


class _Table(object):
    """gather table(s) of one rule book"""
    __slots__ = ("out", "inn", "counts", "vol", "V_out", "V_in", "_host_counts", "flip_ok")

    def __init__(self, out, inn, counts, vol, V_out, V_in, flip_ok=False):
        self.out, self.inn, self.counts, self.vol = out, inn, counts, vol
        self.V_out, self.V_in, self.flip_ok = V_out, V_in, flip_ok
        self._host_counts = None

    def rule_counts(self):
        """per-offset rule counts (host list); one small D2H read, cached"""
        if self._host_counts is None:
            self._host_counts = self.counts.view(self.vol, -1).sum(1).tolist() if self.counts.numel() else \
                [0] * self.vol
        return self._host_counts

    def total_rules(self):
        return float(sum(self.rule_counts()))


class Metadata_3(object):
    """Replaces Metadata<3> (SCN/Metadata/Metadata.h:44-163, pybind class Metadata_3,
    pybind.cpp:12-32,205).  Caches are keyed exactly like the reference's: grids by spatial
    size, submanifold rule books by (spatial, filter), strided rule books by
    (input spatial, filter, stride) (Metadata.cpp:429-443,484-510)."""

    dimension = 3

    def __init__(self):
        self.clear()

    def clear(self):
        self.grids = {}
        self.submanifold = {}
        self.rulebooks = {}
        self.input = None  # dict(point_site, site_off, site_pts, n, V, mode, max_active, spatial)
        self.device = None

    # ---- reference-visible queries ----------------------------------------------------
    def getSpatialLocations(self, spatial_size):
        """LongTensor [V,4] (x,y,z,batch), CPU like the reference (Metadata.cpp:147-168)."""
        return self.getSpatialLocationsDevice(spatial_size).cpu()

    def getSpatialLocationsDevice(self, spatial_size):
        g = self.grids[_key(spatial_size)]
        loc = torch.empty((g.V, 4), dtype=torch.int64, device=g.coords.device)
        check(_hip.load().aabr_spatial_locations(ptr(g.coords), g.V, ptr(loc), stream()))
        return loc

    def getNActive(self, spatial_size):
        return self.grids[_key(spatial_size)].V

    def setInputSpatialSize(self, spatial_size):
        self.input_spatial = _key(spatial_size)

    # ---- builders ---------------------------------------------------------------------------
    def inputLayer(self, spatial_size, coords, batch_size, mode, device):
        """Metadata::inputLayer (Metadata.cpp:405-417)"""
        assert coords.dim() == 2 and coords.size(1) in (3, 4)
        assert self.input is None and len(self.grids) == 0, "Metadata already holds an input layer"
        lib = _hip.load()
        self.device = device
        coords = coords.to(device=device, dtype=torch.int64, non_blocking=True).contiguous()
        n, ncols = coords.shape
        cap = _hip.next_pow2(2 * n)
        nblk = (max(n, 1) + 1023) // 1024
        keys = torch.empty(cap, dtype=torch.int64, device=device)
        vals = torch.empty(cap, dtype=torch.int32, device=device)
        scratch = torch.empty(3 * cap + 2 * n + 4 * nblk + 16, dtype=torch.int32, device=device)
        point_site = torch.empty(n, dtype=torch.int32, device=device)
        site_coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=device)
        site_off = torch.empty(n + 1, dtype=torch.int32, device=device)
        site_pts = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        meta = torch.empty(_hip.META_WORDS, dtype=torch.int32, device=device)
        if n > 0:
            check(lib.aabr_input_layer_sites(ptr(coords), n, ncols, ptr(keys), ptr(vals), cap, ptr(scratch),
                                             ptr(point_site), ptr(site_coords), ptr(site_off), ptr(site_pts),
                                             ptr(meta), stream()))
            m = meta.tolist()  # the one host sync of the input layer: V sizes every later tensor
        else:  # empty scene: an empty grid (all keys EMPTY), nothing to launch
            keys.fill_(-1)
            site_off.zero_()
            m = [0] * _hip.META_WORDS
        if m[2]:
            raise _hip.AabrError("InputLayer: coordinates must lie in [0, 65534] (batch index too)")
        V = m[0]
        key = _key(spatial_size)
        self.grids[key] = _Grid(site_coords[:V], keys, vals, cap, V)
        self.input = dict(point_site=point_site, site_off=site_off, site_pts=site_pts, n=n, V=V,
                          mode=int(mode), max_active=m[1], spatial=key)
        self.input_spatial = key
        return V

    def getSubmanifoldRuleBook(self, spatial_size, filter_size):
        """Metadata::getSubmanifoldRuleBook (Metadata.cpp:429-443) -> cached gather table"""
        k = _key(spatial_size) + _key(filter_size)
        tb = self.submanifold.get(k)
        if tb is None:
            g = self.grids[_key(spatial_size)]
            fs = _key(filter_size)
            vol = fs[0] * fs[1] * fs[2]
            dev = g.keys.device
            table = torch.empty((vol, g.V), dtype=torch.int32, device=dev)
            counts = torch.empty(vol * ((g.V + 255) // 256), dtype=torch.int32, device=dev)
            check(_hip.load().aabr_submanifold_table(ptr(g.coords), g.V, ptr(g.keys), ptr(g.vals), g.cap,
                                                     _hip.i32x3(fs), ptr(table), ptr(counts), stream()))
            # odd filters: the input-gradient gather is the same table read with the mirrored
            # offset (u = v + off_k  <=>  v = u + off_{vol-1-k})
            tb = _Table(table, None, counts, vol, g.V, g.V, flip_ok=all(f % 2 == 1 for f in fs))
            if not tb.flip_ok:
                raise NotImplementedError("even-sized submanifold filters")
            self.submanifold[k] = tb
        return tb

    def getRuleBook(self, in_spatial, out_spatial, filter_size, filter_stride):
        """Metadata::getRuleBook (Metadata.cpp:484-510): builds the output grid on first use."""
        k = _key(in_spatial) + _key(filter_size) + _key(filter_stride)
        tb = self.rulebooks.get(k)
        if tb is None:
            lib = _hip.load()
            gi = self.grids[_key(in_spatial)]
            fs, st, osz = _key(filter_size), _key(filter_stride), _key(out_spatial)
            dev = gi.keys.device
            vol = fs[0] * fs[1] * fs[2]
            maxout = 1
            for a, b in zip(fs, st):
                maxout *= (a + b - 1) // b
            E = gi.V * maxout
            cap = _hip.next_pow2(2 * E)
            nblk = (max(E, 1) + 1023) // 1024
            keys = torch.empty(cap, dtype=torch.int64, device=dev)
            vals = torch.empty(cap, dtype=torch.int32, device=dev)
            scratch = torch.empty(3 * cap + 2 * E + 4 * nblk + 16, dtype=torch.int32, device=dev)
            out_coords = torch.empty((max(E, 1), 4), dtype=torch.int32, device=dev)
            meta = torch.empty(_hip.META_WORDS, dtype=torch.int32, device=dev)
            check(lib.aabr_convolution_sites(ptr(gi.coords), gi.V, _hip.i32x3(fs), _hip.i32x3(st),
                                             _hip.i32x3(osz), ptr(keys), ptr(vals), cap, ptr(scratch),
                                             ptr(out_coords), ptr(meta), stream()))
            V_out = meta.tolist()[0]  # host sync: sizes the output feature tensor
            go = _Grid(out_coords[:V_out], keys, vals, cap, V_out)
            self.grids[osz] = go
            t_out = torch.empty((vol, V_out), dtype=torch.int32, device=dev)
            t_in = torch.empty((vol, gi.V), dtype=torch.int32, device=dev)
            counts = torch.empty(vol * ((V_out + 255) // 256), dtype=torch.int32, device=dev)
            check(lib.aabr_convolution_tables(ptr(gi.coords), gi.V, ptr(gi.keys), ptr(gi.vals), gi.cap,
                                              ptr(go.coords), V_out, ptr(go.keys), ptr(go.vals), go.cap,
                                              _hip.i32x3(fs), _hip.i32x3(st), _hip.i32x3(osz), ptr(t_out),
                                              ptr(t_in), ptr(counts), stream()))
            tb = _Table(t_out, t_in, counts, vol, V_out, gi.V)
            self.rulebooks[k] = tb
        return tb

    # ---- reference-format views (tests / API parity) ------------------------------------------
    def inputLayerRuleBook(self):
        """[[mode, maxActive, nIn, nOut], rules V x (1+maxActive)] (IOLayersRules.h:10-15)"""
        il = self.input
        w = (il["max_active"] if il["mode"] in (3, 4) else 1) + 1
        rules = torch.empty((il["V"], w), dtype=torch.int32, device=il["site_off"].device)
        check(_hip.load().aabr_input_layer_rule_table(ptr(il["site_off"]), ptr(il["site_pts"]), il["V"],
                                                      il["max_active"], il["mode"], ptr(rules), stream()))
        return [il["mode"], w - 1, il["n"], il["V"]], rules

    @staticmethod
    def tableToRuleBook(table):
        """gather table [vol, V] -> list over offsets of int32 [n_k, 2] (entry, row) pairs"""
        vol, V = table.shape
        rules = torch.empty((vol, max(V, 1), 2), dtype=torch.int32, device=table.device)
        counts = torch.empty(vol, dtype=torch.int32, device=table.device)
        check(_hip.load().aabr_table_to_rulebook(ptr(table), V, vol, ptr(rules), ptr(counts), stream()))
        c = counts.tolist()
        return [rules[k, : c[k]] for k in range(vol)]


def _opt(t):
    return t if (t is not None and t.numel() > 0) else None


def _f32c(t, what):
    _hip.require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (the reference path is fp32-only, "
                        "sparseconvnet_cuda.cpp instantiates <float>), got %s" % (what, t.dtype))
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# InputLayer (pybind.cpp:154-162; sparseconvnet_cuda.cpp:432-455)
# ------------------------------------------------------------------------------------------------
def InputLayer_updateOutput(metadata, spatial_size, coords, input_features, output_features, batch_size,
                            mode):
    inp = _f32c(input_features, "input_features")
    mode = int(mode)
    if mode == 0:
        # "guaranteed unique": identical to the summing path when the guarantee holds
        V = metadata.inputLayer(spatial_size, coords, batch_size, 3, inp.device)
        if V != inp.size(0):
            raise _hip.AabrError("InputLayer mode 0 requires unique coordinates")
        metadata.input["mode"] = 3
    else:
        if mode not in (1, 2, 3, 4):
            raise ValueError("InputLayer mode must be 0..4")
        V = metadata.inputLayer(spatial_size, coords, batch_size, mode, inp.device)
    il = metadata.input
    planes = inp.size(1)
    output_features.resize_(V, planes)
    check(_hip.load().aabr_input_layer_forward(ptr(inp), ptr(output_features), V, planes, ptr(il["site_off"]),
                                               ptr(il["site_pts"]), il["mode"], stream()))


def InputLayer_updateGradInput(metadata, d_input_features, d_output_features):
    d_out = _f32c(d_output_features, "d_output_features")
    il = metadata.input
    planes = d_out.size(1)
    d_input_features.resize_(il["n"], planes)
    check(_hip.load().aabr_input_layer_backward(ptr(d_input_features), ptr(d_out), il["n"], planes,
                                                ptr(il["point_site"]), ptr(il["site_off"]),
                                                ptr(il["site_pts"]), il["mode"], stream()))


# ------------------------------------------------------------------------------------------------
# the shared gather-GEMM
# ------------------------------------------------------------------------------------------------
def _conv_fwd(inp, out, n_rows_out, table, vol, weight, bias, flags):
    lib = _hip.load()
    n_in = inp.size(1)
    w = weight.contiguous()
    assert w.size(1) == 1, "groups != 1 is not used by the hot path (FPN_Net never sets it)"
    if flags & 1:
        n_out = w.size(2)
        assert w.size(3) == n_in
    else:
        n_out = w.size(3)
        assert w.size(2) == n_in, (w.shape, n_in)
    out.resize_(n_rows_out, n_out)
    wpack = torch.empty(lib.aabr_conv_wpack_floats(vol, w.size(2), w.size(3)), dtype=torch.float32,
                        device=inp.device)
    check(lib.aabr_conv_forward(ptr(inp), n_in, ptr(out), n_out, n_rows_out, ptr(table), vol, ptr(w),
                                ptr(_opt(bias)), flags, ptr(wpack), stream()))
    return n_out


def _conv_dw(inp, d_out, table, vol, d_weight, d_bias):
    lib = _hip.load()
    n_in, n_out, V_out = inp.size(1), d_out.size(1), d_out.size(0)
    assert d_weight.is_contiguous() and d_weight.numel() == vol * n_in * n_out
    scratch = torch.empty(max(lib.aabr_conv_dw_scratch_floats(V_out, vol, n_in, n_out), 1), dtype=torch.float32,
                          device=inp.device)
    check(lib.aabr_conv_backward_weight(ptr(inp), n_in, ptr(d_out), n_out, V_out, ptr(table), vol,
                                        ptr(d_weight), ptr(_opt(d_bias)), ptr(scratch), stream()))


def _macs(tb, weight):
    if not count_macs:
        return 0.0
    return tb.total_rules() * weight.size(2) * weight.size(3) * weight.size(1)


# SubmanifoldConvolution (pybind.cpp:134-143)
def SubmanifoldConvolution_updateOutput(spatial_size, filter_size, metadata, input_features, output_features,
                                        weight, bias):
    inp = _f32c(input_features, "input_features")
    tb = metadata.getSubmanifoldRuleBook(spatial_size, filter_size)
    _conv_fwd(inp, output_features, tb.V_out, tb.out, tb.vol, weight, bias, 0)
    return _macs(tb, weight)


def SubmanifoldConvolution_backward(spatial_size, filter_size, metadata, input_features, d_input_features,
                                    d_output_features, weight, d_weight, d_bias):
    inp = _f32c(input_features, "input_features")
    d_out = _f32c(d_output_features, "d_output_features")
    tb = metadata.getSubmanifoldRuleBook(spatial_size, filter_size)
    # d_in[u] = sum_k d_out[table[k'][u]] @ W[vol-1-k']^T  (flags: transpose | mirrored offset)
    _conv_fwd(d_out, d_input_features, tb.V_in, tb.out, tb.vol, weight, None, 1 | 2)
    _conv_dw(inp, d_out, tb.out, tb.vol, d_weight, d_bias)


# Convolution (pybind.cpp:54-65)
def Convolution_updateOutput(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                             output_features, weight, bias):
    inp = _f32c(input_features, "input_features")
    tb = metadata.getRuleBook(input_size, output_size, filter_size, filter_stride)
    _conv_fwd(inp, output_features, tb.V_out, tb.out, tb.vol, weight, bias, 0)
    return _macs(tb, weight)


def Convolution_backward(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                         d_input_features, d_output_features, weight, d_weight, d_bias):
    inp = _f32c(input_features, "input_features")
    d_out = _f32c(d_output_features, "d_output_features")
    tb = metadata.getRuleBook(input_size, output_size, filter_size, filter_stride)
    _conv_fwd(d_out, d_input_features, tb.V_in, tb.inn, tb.vol, weight, None, 1)
    _conv_dw(inp, d_out, tb.out, tb.vol, d_weight, d_bias)


# Deconvolution (pybind.cpp:78-89): the rule book is looked up as (outputSize, inputSize) with
# the columns swapped (CPU/Deconvolution.cpp:15-16,34-37)
def Deconvolution_updateOutput(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                               output_features, weight, bias):
    inp = _f32c(input_features, "input_features")
    tb = metadata.getRuleBook(output_size, input_size, filter_size, filter_stride)
    _conv_fwd(inp, output_features, tb.V_in, tb.inn, tb.vol, weight, bias, 0)
    return _macs(tb, weight)


def Deconvolution_backward(input_size, output_size, filter_size, filter_stride, metadata, input_features,
                           d_input_features, d_output_features, weight, d_weight, d_bias):
    inp = _f32c(input_features, "input_features")
    d_out = _f32c(d_output_features, "d_output_features")
    tb = metadata.getRuleBook(output_size, input_size, filter_size, filter_stride)
    _conv_fwd(d_out, d_input_features, tb.V_out, tb.out, tb.vol, weight, None, 1)
    _conv_dw(inp, d_out, tb.inn, tb.vol, d_weight, d_bias)


# ------------------------------------------------------------------------------------------------
# BatchNormalization (pybind.cpp:219-221; batchNormalization.py:120-171)
# ------------------------------------------------------------------------------------------------
def BatchNormalization_updateOutput(input_features, output_features, saveMean, saveInvStd, runningMean,
                                    runningVar, weight, bias, eps, momentum, train, leakiness):
    inp = _f32c(input_features, "input_features")
    lib = _hip.load()
    output_features.resize_as_(inp)
    if inp.dim() != 2:
        return
    rows, planes = inp.shape
    scratch = torch.empty(lib.aabr_bn_scratch_floats(planes), dtype=torch.float32, device=inp.device)
    check(lib.aabr_bn_forward(ptr(inp), ptr(output_features), rows, planes, ptr(saveMean), ptr(saveInvStd),
                              ptr(runningMean), ptr(runningVar), ptr(_opt(weight)), ptr(_opt(bias)), float(eps),
                              float(momentum), int(bool(train)), float(leakiness), ptr(scratch), stream()))


def BatchNormalization_backward(input_features, d_input_features, output_features, d_output_features, saveMean,
                                saveInvStd, runningMean, runningVar, weight, bias, d_weight, d_bias, leakiness):
    """NB: the reference overwrites d_output_features in place with the activation-masked
    gradient (CPU/BatchNormalization.cpp:79-82); nothing downstream reads it, so this
    implementation leaves it untouched (one HBM write pass saved)."""
    inp = _f32c(input_features, "input_features")
    d_out = _f32c(d_output_features, "d_output_features")
    lib = _hip.load()
    d_input_features.resize_as_(inp)
    if inp.dim() != 2:
        return
    rows, planes = inp.shape
    scratch = torch.empty(lib.aabr_bn_scratch_floats(planes), dtype=torch.float32, device=inp.device)
    check(lib.aabr_bn_backward(ptr(inp), ptr(d_input_features), ptr(output_features.contiguous()), ptr(d_out),
                               rows, planes, ptr(saveMean), ptr(saveInvStd), ptr(_opt(weight)),
                               ptr(_opt(d_weight)), ptr(_opt(d_bias)), float(leakiness), ptr(scratch), stream()))

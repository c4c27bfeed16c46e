"""Oracle-side execution of the '2-stage' backbone slice (BASELINE.json configs[1]):
InputLayer(mode 4) -> SubmConv3 C_in->32 -> [Identity | BNReLU -> SubmConv3 -> BNReLU -> SubmConv3] -> add
(reference: SparseConvNet/sparseconvnet/fpn_net.py:42-44,60-75).  Forward and hand-written
backward composed from oracle_lib calls -- test infrastructure only."""
import numpy as np

import oracle_lib as O


def two_stage_forward(locs, feats, W1, W2, W3, bn1, bn2, eps=1e-4, momentum=0.95):
    """bn1/bn2: dict(weight, bias, running_mean, running_var). Returns a cache dict."""
    c = {}
    c["il"] = il = O.input_layer(locs, feats, 4)
    c["rb"] = rb = O.submanifold_rules(il["coords"], [3, 3, 3])
    V = il["V"]
    c["x0"] = il["out"]
    c["x1"], m1 = O.conv_fwd(c["x0"], W1, rb, V)
    c["y1"], c["sm1"], c["si1"], c["rm1"], c["rv1"] = O.bn_fwd(c["x1"], bn1["weight"], bn1["bias"],
                                                             bn1["running_mean"], bn1["running_var"], eps,
                                                             momentum, True, 0.0)
    c["x2"], m2 = O.conv_fwd(c["y1"], W2, rb, V)
    c["y2"], c["sm2"], c["si2"], c["rm2"], c["rv2"] = O.bn_fwd(c["x2"], bn2["weight"], bn2["bias"],
                                                             bn2["running_mean"], bn2["running_var"], eps,
                                                             momentum, True, 0.0)
    c["x3"], m3 = O.conv_fwd(c["y2"], W3, rb, V)
    c["out"] = c["x1"] + c["x3"]
    c["macs"] = m1 + m2 + m3
    return c


def two_stage_backward(c, g_out, W1, W2, W3, bn1, bn2):
    rb = c["rb"]
    d_y2, dW3, _ = O.conv_bwd(c["y2"], g_out, W3, rb)
    d_x2, dbn2w, dbn2b, _ = O.bn_bwd(c["x2"], c["y2"], d_y2, c["sm2"], c["si2"], bn2["weight"], 0.0)
    d_y1, dW2, _ = O.conv_bwd(c["y1"], d_x2, W2, rb)
    d_x1b, dbn1w, dbn1b, _ = O.bn_bwd(c["x1"], c["y1"], d_y1, c["sm1"], c["si1"], bn1["weight"], 0.0)
    d_x1 = g_out + d_x1b
    d_x0, dW1, _ = O.conv_bwd(c["x0"], d_x1, W1, rb)
    d_feats = O.input_layer_bwd(c["il"], d_x0)
    return dict(dW1=dW1, dW2=dW2, dW3=dW3, dbn1w=dbn1w, dbn1b=dbn1b, dbn2w=dbn2w, dbn2b=dbn2b, d_feats=d_feats,
                d_x0=d_x0)


# =====================================================================================================
# Whole FPN_Net on the oracle (reference: SparseConvNet/sparseconvnet/fpn_net.py:40-135 construction,
# :140-152 forward, :168-203 forward_fpn).  A tiny reverse-mode tape records the oracle calls of the
# forward pass and replays their hand-written backward counterparts in reverse order.
# =====================================================================================================
def bf16_round(a):
    """fp32 array rounded to the nearest bf16 value (ties to even), returned as fp32 -- what a bf16 store keeps"""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).reshape(a.shape)
    nan = np.isnan(a)
    if nan.any():
        out = np.where(nan, a, out)
    return out


class _Node(object):
    __slots__ = ("v", "g", "coords", "spatial", "rnd")

    def __init__(self, v, coords, spatial, rnd=None):
        self.v, self.g, self.coords, self.spatial = v, None, coords, tuple(int(s) for s in spatial)
        self.rnd = rnd        # bf16 storage: gradient sums are stored rounded as well

    def add_grad(self, g):
        if self.g is None:
            self.g = g.copy()
        else:
            self.g = self.g + g if self.rnd is None else self.rnd(self.g + g)


class FpnOracle(object):
    """params: the dict returned by `fpn_params(net)`; geometry is cached per scale like Metadata does
    (Metadata.cpp:429-443,484-510)."""

    def __init__(self, params, full_scale, down_kernels, down_strides, rpn_map_sizes, fpn_scales_from_top=(4, 3, 2, 1),
                 roi_scales_from_top=(4, 3, 2, 1), rpn_3d_2d_selector=(1, 2, 3, 4, 5, 6), eps=1e-4, momentum=0.95,
                 leakiness=0.0, train=True, storage="f32"):
        """storage="bf16": the arithmetic model of FPN_Net(feature_dtype=torch.bfloat16) -- an extension the reference
        does not have (it instantiates <float> only, SCN/sparseconvnet_cuda.cpp:281-310).  Everything the device
        STORES as bf16 is rounded here at the same places: the 9->32 input convolution stays fp32 and its output is
        rounded once; from there on every convolution / BatchNorm / add output, every activation gradient and every
        convolution weight (the packed copies the matrix cores read) is a bf16 value; accumulation, BatchNorm
        statistics and all parameter gradients stay fp32 (double here)."""
        assert storage in ("f32", "bf16")
        self.bf16 = storage == "bf16"
        self.rnd = bf16_round if self.bf16 else None
        self.P = params
        self.full_scale = tuple(full_scale)
        self.dk, self.ds = down_kernels, down_strides
        self.rpn_map_sizes = rpn_map_sizes
        self.fst, self.rst, self.sel = list(fpn_scales_from_top), list(roi_scales_from_top), list(rpn_3d_2d_selector)
        self.eps, self.momentum, self.leak, self.train = eps, momentum, leakiness, train
        self.tape = []
        self.grads = {}
        self.macs = 0.0
        self.subm = {}
        self.strided = {}
        self.sites = {}
        self.acts = {}
        self.act_spatial = {}  # name -> spatial size of the level the BatchNorm runs on (tests match rows by coordinates)
        self.override = None   # name -> array: replace a BN output (device activations => identical ReLU masks)
        self.timing = {}

    def _t(self, key, t0):
        import time
        self.timing[key] = self.timing.get(key, 0.0) + time.perf_counter() - t0

    # ---- geometry caches ----
    def _subm_rules(self, node, size):
        k = node.spatial + tuple(size)
        if k not in self.subm:
            import time
            t0 = time.perf_counter()
            self.subm[k] = O.submanifold_rules(node.coords, list(size))
            self._t("rule_build", t0)
        return self.subm[k]

    def _strided_rules(self, coords, in_spatial, size, stride):
        k = tuple(in_spatial) + tuple(size) + tuple(stride)
        if k not in self.strided:
            import time
            t0 = time.perf_counter()
            osz = tuple((np.array(in_spatial) - np.array(size)) // np.array(stride) + 1)
            rb, oc = O.convolution_rules(coords, list(size), list(stride), list(osz))
            self.strided[k] = (rb, oc, osz)
            self.sites[osz] = oc
            self._t("rule_build", t0)
        return self.strided[k]

    # ---- ops ----
    def input_layer(self, locs, feats):
        import time
        t0 = time.perf_counter()
        il = O.input_layer(locs, feats, 4)
        self._t("scatter", t0)
        self.il = il
        n = _Node(il["out"], il["coords"], self.full_scale)
        self.sites[self.full_scale] = il["coords"]

        def bwd():
            if n.g is not None:
                self.grads["d_feats"] = O.input_layer_bwd(il, n.g)
        self.tape.append(bwd)
        return n

    def subm_conv(self, x, name, size):
        import time
        W0 = self.P[name]
        q = self.bf16 and name != "layers_in.1"
        W = bf16_round(W0) if q else W0
        rb = self._subm_rules(x, size)
        t0 = time.perf_counter()
        out, m = O.conv_fwd(x.v, W.reshape(rb.vol, x.v.shape[1], -1), rb, x.v.shape[0])
        self._t("conv", t0)
        self.macs += m
        y = _Node(bf16_round(out) if q else out, x.coords, x.spatial, self.rnd if q else None)

        def bwd():
            if y.g is None:
                return
            t0 = time.perf_counter()
            d_in, dW, _ = O.conv_bwd(x.v, y.g, W.reshape(rb.vol, x.v.shape[1], -1), rb)
            self._t("conv", t0)
            self.grads[name] = dW.reshape(W0.shape)
            x.add_grad(bf16_round(d_in) if q else d_in)
        self.tape.append(bwd)
        return y

    def to_storage(self, x):
        """the cast behind the input convolution (FPN_Net._cast / the compiled graph's first record)"""
        if not self.bf16:
            return x
        y = _Node(bf16_round(x.v), x.coords, x.spatial, self.rnd)

        def bwd():
            if y.g is not None:
                x.add_grad(y.g)          # bf16 -> fp32: exact
        self.tape.append(bwd)
        return y

    def conv(self, x, name, size, stride):
        import time
        W0 = self.P[name]
        W = bf16_round(W0) if self.bf16 else W0
        rb, oc, osz = self._strided_rules(x.coords, x.spatial, size, stride)
        t0 = time.perf_counter()
        out, m = O.conv_fwd(x.v, W.reshape(rb.vol, x.v.shape[1], -1), rb, oc.shape[0])
        self._t("conv", t0)
        self.macs += m
        y = _Node(bf16_round(out) if self.bf16 else out, oc, osz, self.rnd)

        def bwd():
            if y.g is None:
                return
            t0 = time.perf_counter()
            d_in, dW, _ = O.conv_bwd(x.v, y.g, W.reshape(rb.vol, x.v.shape[1], -1), rb)
            self._t("conv", t0)
            self.grads[name] = dW.reshape(W0.shape)
            x.add_grad(bf16_round(d_in) if self.bf16 else d_in)
        self.tape.append(bwd)
        return y

    def deconv(self, x, name, size, stride, fine_spatial):
        """CPU/Deconvolution.cpp:15-16: rule book of the (fine -> coarse) convolution, columns swapped"""
        import time
        W0 = self.P[name]
        W = bf16_round(W0) if self.bf16 else W0
        rb, oc, osz = self._strided_rules(self.sites[tuple(fine_spatial)], fine_spatial, size, stride)
        assert osz == x.spatial and oc.shape[0] == x.v.shape[0]
        fine = self.sites[tuple(fine_spatial)]
        t0 = time.perf_counter()
        out, m = O.conv_fwd(x.v, W.reshape(rb.vol, x.v.shape[1], -1), rb, fine.shape[0], in_col=1)
        self._t("conv", t0)
        self.macs += m
        y = _Node(bf16_round(out) if self.bf16 else out, fine, fine_spatial, self.rnd)

        def bwd():
            if y.g is None:
                return
            t0 = time.perf_counter()
            d_in, dW, _ = O.conv_bwd(x.v, y.g, W.reshape(rb.vol, x.v.shape[1], -1), rb, in_col=1)
            self._t("conv", t0)
            self.grads[name] = dW.reshape(W0.shape)
            x.add_grad(bf16_round(d_in) if self.bf16 else d_in)
        self.tape.append(bwd)
        return y

    def bn_relu(self, x, name):
        import time
        p = self.P[name]
        t0 = time.perf_counter()
        out, sm, si, rm, rv = O.bn_fwd(x.v, p["weight"], p["bias"], p["running_mean"], p["running_var"], self.eps,
                                       self.momentum, self.train, self.leak)
        self._t("bn", t0)
        if self.bf16:
            out = bf16_round(out)
        self.acts[name] = out
        self.act_spatial[name] = x.spatial
        if self.override is not None and name in self.override:
            out = self.override[name]
        p["running_mean_out"], p["running_var_out"] = rm, rv
        y = _Node(out, x.coords, x.spatial, self.rnd)

        def bwd():
            if y.g is None:
                return
            t0 = time.perf_counter()
            d_in, dw, db, _ = O.bn_bwd(x.v, y.v, y.g, sm, si, p["weight"], self.leak)
            self._t("bn", t0)
            self.grads[name + ".weight"], self.grads[name + ".bias"] = dw, db
            x.add_grad(bf16_round(d_in) if self.bf16 else d_in)
        self.tape.append(bwd)
        return y

    def add(self, a, b):
        y = _Node(bf16_round(a.v + b.v) if self.bf16 else a.v + b.v, a.coords, a.spatial, self.rnd)

        def bwd():
            if y.g is not None:
                a.add_grad(y.g)
                b.add_grad(y.g)
        self.tape.append(bwd)
        return y

    # ---- the network ----
    def forward(self, locs, feats):
        nscale = 1 + len(self.dk)
        net = self.to_storage(self.subm_conv(self.input_layer(locs, feats), "layers_in.1", (3, 3, 3)))
        downs = []
        for k in range(nscale):
            if k > 0:
                net = self.conv(self.bn_relu(net, "m_downs.%d.down.bn" % k), "m_downs.%d.down.conv" % k,
                                self.dk[k - 1], self.ds[k - 1])
            r = 0
            while "m_downs.%d.block%d.bn1" % (k, r) in self.P:
                b = "m_downs.%d.block%d." % (k, r)
                h = self.subm_conv(self.bn_relu(net, b + "bn1"), b + "conv1", (3, 3, 3))
                h = self.subm_conv(self.bn_relu(h, b + "bn2"), b + "conv2", (3, 3, 3))
                net = self.add(net, h)
                r += 1
            downs.append(net)
        net = self.subm_conv(net, "m_shortcuts.%d" % (nscale - 1), (1, 1, 1))
        ups = [net]
        for k in range(nscale - 1):
            j = nscale - 2 - k
            net = self.deconv(self.bn_relu(net, "m_ups.%d.bn" % k), "m_ups.%d.deconv" % k, self.dk[j], self.ds[j],
                              downs[j].spatial)
            net = self.add(net, self.subm_conv(downs[j], "m_shortcuts.%d" % j, (1, 1, 1)))
            ups.append(self.subm_conv(net, "m_mergeds.%d" % k, (3, 3, 3)))
        rpn3d = [ups[i] for i in self.fst]
        rpn2d = [self.conv(rpn3d[i], "convs_pro2d.%d" % i, (1, 1, int(self.rpn_map_sizes[i][2])), (1, 1, 1))
                 for i in range(len(rpn3d))]
        for i, mp in enumerate(rpn3d):
            assert mp.spatial == tuple(int(v) for v in self.rpn_map_sizes[i])
        allm = rpn3d + rpn2d
        self.rpn_maps = [allm[i] for i in self.sel]
        self.roi_maps = [ups[i] for i in self.rst]
        return self.rpn_maps, self.roi_maps

    def backward(self, rpn_grads):
        """rpn_grads: list of arrays (or None) matching self.rpn_maps"""
        for mp, g in zip(self.rpn_maps, rpn_grads):
            if g is not None:      # bf16 storage: the maps are handed out as fp32 copies, their gradient is stored bf16
                g = np.ascontiguousarray(g, np.float32)
                mp.add_grad(bf16_round(g) if self.bf16 else g)
        for fn in reversed(self.tape):
            fn()
        return self.grads


def fpn_params(net):
    """numpy parameters of a (repo or reference-shaped) FPN_Net module tree, keyed for FpnOracle"""
    P = {}
    n = lambda t: t.detach().float().cpu().numpy().copy()

    def bn(m):
        return dict(weight=n(m.weight), bias=n(m.bias), running_mean=n(m.running_mean), running_var=n(m.running_var))

    P["layers_in.1"] = n(net.layers_in[1].weight)
    for k, m in enumerate(net.m_downs):
        mods = list(m.children())
        i = 0
        if k > 0:
            dn = list(mods[0].children())
            P["m_downs.%d.down.bn" % k], P["m_downs.%d.down.conv" % k] = bn(dn[0]), n(dn[1].weight)
            i = 1
        r = 0
        while i < len(mods):
            seq = list(list(mods[i].children())[1].children())   # ConcatTable: [Identity, Sequential(...)]
            b = "m_downs.%d.block%d." % (k, r)
            P[b + "bn1"], P[b + "conv1"], P[b + "bn2"], P[b + "conv2"] = bn(seq[0]), n(seq[1].weight), bn(seq[2]), \
                n(seq[3].weight)
            i += 2   # ConcatTable, AddTable
            r += 1
    for k, m in enumerate(net.m_shortcuts):
        P["m_shortcuts.%d" % k] = n(m.weight)
    for k, m in enumerate(net.m_ups):
        mods = list(m.children())
        P["m_ups.%d.bn" % k], P["m_ups.%d.deconv" % k] = bn(mods[0]), n(mods[1].weight)
    for k, m in enumerate(net.m_mergeds):
        P["m_mergeds.%d" % k] = n(m.weight)
    for k, m in enumerate(net.convs_pro2d):
        P["convs_pro2d.%d" % k] = n(m.weight)
    return P


def fpn_param_names(net):
    """FpnOracle gradient key -> torch parameter (same walk as fpn_params)"""
    M = {}
    M["layers_in.1"] = net.layers_in[1].weight
    for k, m in enumerate(net.m_downs):
        mods = list(m.children())
        i = 0
        if k > 0:
            dn = list(mods[0].children())
            M["m_downs.%d.down.bn.weight" % k], M["m_downs.%d.down.bn.bias" % k] = dn[0].weight, dn[0].bias
            M["m_downs.%d.down.conv" % k] = dn[1].weight
            i = 1
        r = 0
        while i < len(mods):
            seq = list(list(mods[i].children())[1].children())
            b = "m_downs.%d.block%d." % (k, r)
            M[b + "bn1.weight"], M[b + "bn1.bias"], M[b + "conv1"] = seq[0].weight, seq[0].bias, seq[1].weight
            M[b + "bn2.weight"], M[b + "bn2.bias"], M[b + "conv2"] = seq[2].weight, seq[2].bias, seq[3].weight
            i += 2
            r += 1
    for k, m in enumerate(net.m_shortcuts):
        M["m_shortcuts.%d" % k] = m.weight
    for k, m in enumerate(net.m_ups):
        mods = list(m.children())
        M["m_ups.%d.bn.weight" % k], M["m_ups.%d.bn.bias" % k], M["m_ups.%d.deconv" % k] = mods[0].weight, \
            mods[0].bias, mods[1].weight
    for k, m in enumerate(net.m_mergeds):
        M["m_mergeds.%d" % k] = m.weight
    for k, m in enumerate(net.convs_pro2d):
        M["convs_pro2d.%d" % k] = m.weight
    return M


def fpn_bn_modules(net):
    """FpnOracle BN name -> torch BatchNorm module (same walk as fpn_params)"""
    M = {}
    for k, m in enumerate(net.m_downs):
        mods = list(m.children())
        i = 0
        if k > 0:
            M["m_downs.%d.down.bn" % k] = list(mods[0].children())[0]
            i = 1
        r = 0
        while i < len(mods):
            seq = list(list(mods[i].children())[1].children())
            M["m_downs.%d.block%d.bn1" % (k, r)], M["m_downs.%d.block%d.bn2" % (k, r)] = seq[0], seq[2]
            i += 2
            r += 1
    for k, m in enumerate(net.m_ups):
        M["m_ups.%d.bn" % k] = list(m.children())[0]
    return M

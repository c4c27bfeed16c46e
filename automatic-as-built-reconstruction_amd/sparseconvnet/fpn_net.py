"""FPN_Net -- the sparse ResNet-FPN backbone of the detector.

Same 17-argument constructor, sub-module names (layers_in, m_downs, m_shortcuts, m_ups,
m_mergeds, convs_pro2d ...) and forward contract as the reference
(SparseConvNet/sparseconvnet/fpn_net.py:13-203), so reference checkpoints load by name
(maskrcnn_benchmark/utils/checkpoint.py:105-106) and `SparseRCNN.forward`
(modeling/detector/sparse_rcnn.py:53) can call it unchanged.  Debug printing and pdb hooks
of the reference are not reproduced."""
import numpy as np
import torch
import torch.nn as nn

import sparseconvnet as scn


class FPN_Net(torch.nn.Module):
    def __init__(self, full_scale, dimension, raw_elements, reps, nPlanesF, nPlaneM, residual_blocks,
                 fpn_scales_from_top, roi_scales_from_top, downsample, rpn_map_sizes, rpn_3d_2d_selector,
                 leakiness=0, voxel_scale=None, bn_momentum=0.9, track_running_stats=True,
                 feature_dtype=torch.float32):
        """`feature_dtype` (extension, not in the reference): torch.bfloat16 stores every feature
        matrix after the first (raw_elements -> nPlanesF[0]) convolution in bf16; parameters, batch-norm
        statistics and all accumulation stay fp32, the returned maps are cast back to fp32."""
        nn.Module.__init__(self)
        assert feature_dtype in (torch.float32, torch.bfloat16)
        self.feature_dtype = feature_dtype
        self.prebuild_geometry = True   # extension: see _prebuild_geometry
        self.prepack_weights = True     # extension: see _refresh_weight_packs
        self.grids_from_input = True    # extension: see _grids_from_input
        self.compiled_graph = False     # extension: planExecutor.run_fpn (one launch list per pass)
        self.site_order = "first_seen"  # extension: "brick" = brick-major rows over brick grids (see set_site_order)
        # extension (opt-in, never the default): the reference's forward_fpn runs the whole top-down path to scale 0
        # (fpn_net.py:181-196; the `continue` that would stop it is commented out) although only ups[i], i in
        # fpn_scales_from_top + roi_scales_from_top, are returned.  True: stop the top-down path behind the last consumed
        # level -- m_ups / m_shortcuts / m_mergeds below it are not run (and get no gradient, as before: nothing returned
        # depends on them).  Returned maps are bit-identical; forward_pass_multiplyAdd_count is smaller by their MACs.
        self.prune_unused_levels = False
        self.bn_momentum = bn_momentum
        self.track_running_stats = track_running_stats
        self.dimension = dimension
        self.down_kernels = downsample[0]
        self.down_strides = downsample[1]
        self.fpn_scales_from_top = fpn_scales_from_top
        self.roi_scales_from_top = roi_scales_from_top
        scale_num = len(nPlanesF)
        assert len(self.down_kernels) == scale_num - 1 == len(self.down_strides)
        assert all(len(ks) == 3 for ks in self.down_kernels)
        assert all(len(ss) == 3 for ss in self.down_strides)
        self._merge = "add"

        ele_channels = {"xyz": 3, "color": 3, "normal": 3}
        in_channels = sum(ele_channels[e] for e in raw_elements)
        bn = dict(momentum=bn_momentum, track_running_stats=track_running_stats)

        self.layers_in_0 = scn.Sequential(scn.InputLayer(dimension, full_scale, mode=4))
        self.layers_in = scn.Sequential(
            scn.InputLayer(dimension, full_scale, mode=4),
            scn.SubmanifoldConvolution(dimension, in_channels, nPlanesF[0], 3, False))
        # the reference also owns layers_out (BatchNormReLU + OutputLayer) and a 20-way linear
        # head it never calls in forward (fpn_net.py:46-50); kept by name
        self.layers_out = scn.Sequential(scn.BatchNormReLU(nPlanesF[0], **bn), scn.OutputLayer(dimension))
        self.linear = nn.Linear(nPlanesF[0], 20)
        self.voxel_scale = voxel_scale
        self.rpn_map_sizes = np.array(rpn_map_sizes)
        self.rpn_3d_2d_selector = rpn_3d_2d_selector

        self.convs_pro2d = nn.ModuleList()
        for zsize in self.rpn_map_sizes[:, -1]:
            self.convs_pro2d.append(scn.Convolution(self.dimension, nPlaneM, nPlaneM, [1, 1, int(zsize)],
                                                    [1, 1, 1], False))

        def block(m, a, b):
            if residual_blocks:  # ResNet style
                m.add(scn.ConcatTable()
                      .add(scn.Identity() if a == b else scn.NetworkInNetwork(a, b, False))
                      .add(scn.Sequential()
                           .add(scn.BatchNormLeakyReLU(a, leakiness=leakiness, **bn))
                           .add(scn.SubmanifoldConvolution(dimension, a, b, 3, False))
                           .add(scn.BatchNormLeakyReLU(b, leakiness=leakiness, **bn))
                           .add(scn.SubmanifoldConvolution(dimension, b, b, 3, False)))
                      ).add(scn.AddTable())
            else:  # VGG style
                m.add(scn.Sequential()
                      .add(scn.BatchNormLeakyReLU(a, leakiness=leakiness, **bn))
                      .add(scn.SubmanifoldConvolution(dimension, a, b, 3, False)))
            return {"kernel": [1, 1, 1], "stride": [1, 1, 1]}

        def down(m, nPlane_in, nPlane_downed, scale):
            m.add(scn.Sequential()
                  .add(scn.BatchNormLeakyReLU(nPlane_in, leakiness=leakiness, **bn))
                  .add(scn.Convolution(dimension, nPlane_in, nPlane_downed, self.down_kernels[scale],
                                       self.down_strides[scale], False)))
            return {"kernel": self.down_kernels[scale], "stride": self.down_strides[scale]}

        def up(m, nPlane_in, nPlane_uped, scale):
            m.add(scn.BatchNormLeakyReLU(nPlane_in, leakiness=leakiness, **bn)).add(
                scn.Deconvolution(dimension, nPlane_in, nPlane_uped, self.down_kernels[scale],
                                  self.down_strides[scale], False))
            return {"kernel": self.down_kernels[scale], "stride": self.down_strides[scale]}

        scales_num = len(nPlanesF)
        m_downs = nn.ModuleList()
        m_shortcuts = nn.ModuleList()
        operations_down = []
        for k in range(scales_num):
            m = scn.Sequential()
            if k > 0:
                operations_down.append(down(m, nPlanesF[k - 1], nPlanesF[k], k - 1))
            for _ in range(reps):
                op = block(m, nPlanesF[k], nPlanesF[k])
                if k == 0:
                    operations_down.append(op)
            m_downs.append(m)
            m_shortcuts.append(scn.SubmanifoldConvolution(dimension, nPlanesF[k], nPlaneM, 1, False))

        m_ups = nn.ModuleList()
        m_mergeds = nn.ModuleList()
        operations_up = []
        for k in range(scales_num - 1, 0, -1):
            m = scn.Sequential()
            operations_up.append(up(m, nPlaneM, nPlaneM, k - 1))
            m_ups.append(m)
            m_mergeds.append(scn.SubmanifoldConvolution(dimension, nPlaneM, nPlaneM, 3, False))

        self.m_downs = m_downs
        self.m_shortcuts = m_shortcuts
        self.m_ups = m_ups
        self.m_mergeds = m_mergeds
        self.operations_down = operations_down
        self.operations_up = operations_up

    def set_site_order(self, order):
        """Extension: "brick" numbers the sites of every level brick by brick (SCN.Metadata_3(site_order="brick"),
        csrc/brick.hip) instead of in the reference's first-seen / insertion order: same sites, same features per site,
        rows of a sample permuted (the returned maps come with their own `get_spatial_locations`, as always); spatial
        neighbours are neighbours in memory and no hash table is built or probed."""
        assert order in ("first_seen", "brick")
        self.site_order = order
        self.layers_in[0].site_order = order
        self.layers_in_0[0].site_order = order
        return self

    @staticmethod
    def _cast(net, dtype):
        if net.features.dtype == dtype:
            return net
        out = scn.SparseConvNetTensor()
        out.metadata = net.metadata
        out.spatial_size = net.spatial_size
        out.features = net.features.to(dtype)
        return out

    def _prebuild_geometry(self, net):
        """Extension (not in the reference): build every grid and rule table of the pass NOW, before the first
        convolution is enqueued.  The geometry depends on the coordinates only; each strided level costs one small
        device->host read (its site count sizes the next tensors), and a read drains the stream -- so the reads are
        taken here, back to back on a nearly empty queue, instead of one in front of every down-sampling layer where
        each would wait for all the convolutions queued before it and leave the GPU idle while the host catches up.
        Same rule books, same cache keys (Metadata caches by (spatial, filter[, stride])): the layers find them."""
        from . import SCN
        md = net.metadata
        if getattr(md, "_fpn_prebuilt", None) is not None:
            return md._fpn_prebuilt
        sp = self._size_plan(net.spatial_size)
        sizes = sp["sizes"]
        three, one = sp["three"], sp["one"]
        nscale = len(self.m_downs)
        if self.grids_from_input:
            self._grids_from_input(md, sp)
        for k in range(nscale):
            sz = sizes[k]
            md.getSubmanifoldRuleBook(sz, three)
            if (nscale - 1 - k) <= self._top_down_levels():      # (the lateral 1x1x1 of a level the top-down path reaches)
                md.getSubmanifoldRuleBook(sz, one)
            if k + 1 < nscale:
                ks, st = sp["down"][k]
                md.getRuleBook(sz, sizes[k + 1], ks, st)
        for msz, out, ks in sp["rpn"]:
            md.getRuleBook(msz, out, ks, one)
        if SCN.count_macs:      # rule totals of all books of the pass (the MAC counter's terms): one launch
            SCN.prefetch_totals([tb.out for tb in list(md.submanifold.values()) + list(md.rulebooks.values())])
        md._fpn_prebuilt = sizes
        return sizes

    def _size_plan(self, sz0):
        """The spatial sizes of every level of the pass, the filter / stride tensors and the grid specifications of
        `_grids_from_input`: functions of the input's spatial size and the constructor arguments only.  Computed once
        per input size and kept as long-lived tensor objects (SCN._key memoises their tuple form per object), instead
        of ~150 small LongTensor constructions, divisions and .tolist() calls per pass (1 ms of host time)."""
        from . import SCN
        plans = self.__dict__.setdefault("_size_plans", {})
        k0 = SCN._key(sz0)
        sp = plans.get(k0)
        if sp is not None:
            return sp
        nscale = len(self.m_downs)
        three, one = torch.LongTensor([3, 3, 3]), torch.LongTensor([1, 1, 1])
        sz = torch.LongTensor(list(k0))
        sizes, down = [sz], []
        for k in range(nscale - 1):
            ks, st = torch.LongTensor(self.down_kernels[k]), torch.LongTensor(self.down_strides[k])
            sz = (sz - ks) // st + 1
            down.append((ks, st))
            sizes.append(sz)
        rpn = []
        for i, scale_from_top in enumerate(self.fpn_scales_from_top):
            msz = sizes[nscale - 1 - scale_from_top]
            ks = torch.LongTensor([1, 1, int(self.rpn_map_sizes[i][2])])
            rpn.append((msz, (msz - ks) // one + 1, ks))
        sp = dict(sizes=sizes, three=three, one=one, down=down, rpn=rpn, rounds=self._grid_rounds(sizes, down))
        plans[k0] = sp
        return sp

    def _grid_rounds(self, sizes, down):
        """`_grids_from_input`'s rounds for one input size: [(base size, base key, [(out size, out key, cumulative
        stride relative to the base)])], or None when a down-sampling level overlaps (filter != stride)"""
        nscale = len(self.m_downs)
        cum = torch.LongTensor([1, 1, 1])
        lvl = [(sizes[0], cum)]
        for k in range(nscale - 1):
            ks, st = down[k]
            if not torch.equal(ks, st):
                return None
            cum = cum * st
            lvl.append((sizes[k + 1], cum))
        zmaps = {}                                   # level -> z-collapse filter that spans the whole z extent
        for i, scale_from_top in enumerate(self.fpn_scales_from_top):
            k = nscale - 1 - scale_from_top
            z = int(self.rpn_map_sizes[i][2])
            if z == int(lvl[k][0][2]) and z > 1:
                zmaps[k] = z
        step, rounds = 4, []
        for base in range(0, nscale, step):
            bsz, bcum = lvl[base]
            specs = []
            for k in range(base + 1, min(base + step, nscale - 1) + 1):
                specs.append((lvl[k][0], lvl[k][1] // bcum))
            for k, z in zmaps.items():
                if base <= k < base + step:
                    # a z-collapse grid needs its level's sites only through their (x, y): from the base as well
                    specs.append((torch.LongTensor([int(lvl[k][0][0]), int(lvl[k][0][1]), 1]),
                                  (lvl[k][1] // bcum) * torch.LongTensor([1, 1, z])))
            specs = [(o, tuple(int(v) for v in o.tolist()), c) for o, c in specs if int(c.max()) <= 65536]
            rounds.append((bsz, tuple(int(v) for v in bsz.tolist()), specs))
        return rounds

    def _brick_specs(self, sp):
        """[(out key, source key, size, stride)] of the pass's strided levels in dependency order (cached per size plan)"""
        specs = sp.get("brick_specs")
        if specs is None:
            from . import SCN
            key = SCN._key
            specs = []
            nscale = len(self.m_downs)
            for k in range(nscale - 1):
                ks, st = sp["down"][k]
                specs.append((key(sp["sizes"][k + 1]), key(sp["sizes"][k]), key(ks), key(st)))
            for msz, out, ks in sp["rpn"]:
                # a [1, 1, z] / stride-1 filter that spans the whole z extent: its output sites are the level's (x, y)
                # columns = the non-overlapping form (1, 1, z) / (1, 1, z); otherwise the overlapping form as it is
                if key(out) != key(msz):
                    specs.append((key(out), key(msz), key(ks), key(ks) if key(out)[2] == 1 else (1, 1, 1)))
            sp["brick_specs"] = specs
        return specs

    def _grids_from_input(self, md, sp):
        """the strided grids of the pass in rounds of four levels when the down-sampling levels do not overlap
        (filter == stride, the reference's default [[2,2,2]]*8): every grid of a round is built straight from the
        round's base grid (Metadata.buildGridsFromInput) and the round costs ONE host read -- 3 reads per pass
        instead of 13.  Not all from level 0: a 49-site level built from 310 k sites is 310 k atomics on 49 words
        (measured: +2 ms); four levels deep the contention stays below ~20 per word on scene data.  The rounds
        themselves are part of the size plan (`_grid_rounds`)."""
        from . import SCN
        g0 = md.grids.get(SCN._key(sp["sizes"][0]))
        if g0 is not None and g0.brick is not None:
            # brick grids: every level from its PARENT level with device-side counts, the z-collapse grids from their own
            # level -- ONE host read for the whole pyramid (Metadata_3.buildBrickPyramid)
            md.buildBrickPyramid(self._brick_specs(sp), defer=bool(getattr(md, "_defer_pyramid", False)))
            return
        if sp["rounds"] is None:
            return
        for bsz, bkey, specs in sp["rounds"]:
            if bkey not in md.grids:
                return
            todo = [(o, c) for o, okey, c in specs if okey not in md.grids]
            md.buildGridsFromInput(bsz, todo)

    def _compile_streams(self, md, sizes, in_channels):
        """block streams / offset-pair lists of every convolution of the pass (forward, input-gradient and weight-
        gradient launches), built now instead of at each layer's first use"""
        from . import SCN
        dt = self.feature_dtype
        f32 = torch.float32
        three, one = (3, 3, 3), (1, 1, 1)
        nscale = len(self.m_downs)
        key = SCN._key            # memoised per (long-lived) size tensor of the size plan

        def subm(k, fs, ci, co, dtype):
            tb = md.submanifold[key(sizes[k]) + fs]
            SCN.compile_streams(tb.out, tb.V_in, ci, co, dtype, weight_grad=True)    # forward + dW
            SCN.compile_streams(tb.out, tb.V_out, co, ci, dtype)                      # input gradient (mirrored)

        def strided(tb, ci, co, dtype, transposed):
            if not transposed:     # Convolution: fine -> coarse
                SCN.compile_streams(tb.out, tb.V_in, ci, co, dtype, weight_grad=True)
                SCN.compile_streams(tb.inn, tb.V_out, co, ci, dtype)
            else:                  # Deconvolution over the same book: coarse -> fine
                SCN.compile_streams(tb.inn, tb.V_out, ci, co, dtype, weight_grad=True)
                SCN.compile_streams(tb.out, tb.V_in, co, ci, dtype)

        planes = [self.m_shortcuts[k].nIn for k in range(nscale)]
        nM = self.m_shortcuts[0].nOut
        live = lambda k: (nscale - 1 - k) <= self._top_down_levels()   # the top-down path reaches scale k
        subm(0, three, in_channels, planes[0], f32)               # layers_in convolution runs in fp32
        for k in range(nscale):
            subm(k, three, planes[k], planes[k], dt)              # residual blocks
            if live(k):
                subm(k, one, planes[k], nM, dt)                   # lateral 1x1x1
            if k + 1 < nscale:
                ks, st = tuple(self.down_kernels[k]), tuple(self.down_strides[k])
                tb = md.rulebooks[key(sizes[k]) + ks + st]
                strided(tb, planes[k], planes[k + 1], dt, False)  # down-sampling convolution
                if live(k):
                    strided(tb, nM, nM, dt, True)                 # up-sampling deconvolution of the same book
            if k < nscale - 1 and live(k):
                subm(k, three, nM, nM, dt)                        # merged 3x3x3 on the up path
        for i, scale_from_top in enumerate(self.fpn_scales_from_top):
            msz = sizes[nscale - 1 - scale_from_top]
            ks = (1, 1, int(self.rpn_map_sizes[i][2]))
            strided(md.rulebooks[key(msz) + ks + one], nM, nM, dt, False)

    def prepare(self, input, stream):
        """Extension (not in the reference): build the NEXT batch's whole geometry -- voxel grid, every strided grid,
        rule tables, compiled block streams -- on `stream` while the current batch still trains on the main stream.
        The geometry depends on the coordinates only, so this is the device-side analogue of a data-loader prefetch;
        its host reads (one site count per grid) wait for `stream` alone.  `forward` picks the prepared Metadata up
        when it is called with the same coordinate tensor (InputLayer._prepared).
        = `prepare_begin` + `prepare_end`: a caller with other launches to enqueue puts them between the two, and the host
        then does not sit waiting while the device builds the level pyramid (brick grids only)."""
        self.prepare_begin(input, stream)
        self.prepare_end()

    def prepare_begin(self, input, stream):
        """first half of `prepare`: voxel scatter, (brick grids:) the input level's bricks and the whole pyramid of strided
        levels ENQUEUED, their counts' read posted"""
        coords = input[0]
        inp_layer = self.layers_in[0]
        dev = coords.device if coords.is_cuda else torch.device("cuda", torch.cuda.current_device())
        inp_layer.prepare(coords, dev, stream)
        src, ver, c64, md = inp_layer._prepared[-1]
        from . import SCN
        SCN._reap_handed_over()      # geometry of earlier passes whose parking event has passed (also reaped by forward)
        with torch.cuda.stream(stream):
            md.inputLayerFinish()
            g0 = md.grids.get(SCN._key(inp_layer.spatial_size))
            if g0 is not None and g0.brick is not None and self.grids_from_input:
                sp = self._size_plan(inp_layer.spatial_size)
                md._defer_pyramid = True
                try:
                    with SCN.geom_plan():
                        self._grids_from_input(md, sp)
                finally:
                    md._defer_pyramid = False
        self._preparing = (md, stream)

    def prepare_end(self):
        """second half of `prepare`: collect the pyramid's counts, then every rule table and block stream"""
        md, stream = self.__dict__.pop("_preparing")
        inp_layer = self.layers_in[0]
        from . import SCN
        with torch.cuda.stream(stream):
            md.finishBrickPyramid()
            stub = scn.SparseConvNetTensor(None, md, inp_layer.spatial_size)
            with SCN.geom_plan():     # the builders' launches as a few lists (one per blocking read) instead of ~250 calls
                sizes = self._prebuild_geometry(stub)
                in_channels = self.layers_in[1].nIn
                self._compile_streams(md, sizes, in_channels)
        md.prepared_on = stream

    def prefetcher(self):
        """Extension: the helper thread that runs `prepare` (see GeometryPrefetcher); one per network"""
        pf = getattr(self, "_prefetcher", None)
        if pf is None:
            pf = self._prefetcher = GeometryPrefetcher(self)
        return pf

    def _refresh_weight_packs(self):
        """Extension: all convolution weights packed (both orientations) by one launch per weight version"""
        from . import SCN
        plan = getattr(self, "_pack_plan", None)
        if plan is None or plan.dtype != self.feature_dtype:
            ws = [m.weight for m in self.modules()
                  if isinstance(m, (scn.SubmanifoldConvolution, scn.Convolution, scn.Deconvolution))]
            plan = self._pack_plan = SCN.WeightPackPlan(ws, self.feature_dtype)
        plan.refresh()
        return plan

    def forward(self, net0):
        plan = self._refresh_weight_packs() if self.prepack_weights else None
        try:
            return self._forward(net0)
        finally:
            if plan is not None:
                plan.release()

    def _forward(self, net0):
        net1 = self.layers_in(net0)
        if self.prebuild_geometry:
            from . import SCN
            with SCN.geom_plan():
                self._prebuild_geometry(net1)
        if self.compiled_graph and self.prepack_weights:
            from . import planExecutor
            out = planExecutor.run_fpn(self, net1)
            if out is not None:
                return out
        if self.feature_dtype == torch.float32:
            return self.forward_fpn(net1)
        rpn_maps, roi_maps = self.forward_fpn(self._cast(net1, self.feature_dtype))
        return ([self._cast(m, torch.float32) for m in rpn_maps], [self._cast(m, torch.float32) for m in roi_maps])

    def _top_down_levels(self):
        """m_ups / m_mergeds stages forward_fpn runs: all of them (the reference), or up to the last consumed map"""
        n = len(self.m_downs) - 1
        if not self.prune_unused_levels:
            return n
        return min(n, max(list(self.fpn_scales_from_top) + list(self.roi_scales_from_top)))

    def forward_fpn(self, net):
        scales_num = len(self.m_downs)
        downs = []
        for m in self.m_downs:
            net = m(net)
            downs.append(net)
        net = self.m_shortcuts[-1](net)
        ups = [net]
        for k in range(self._top_down_levels()):
            j = scales_num - 1 - k - 1
            net = self.m_ups[k](net)
            shorcut = self.m_shortcuts[j](downs[j])
            net = scn.add_feature_planes([net, shorcut])
            ups.append(self.m_mergeds[k](net))

        rpn_maps_3d = [ups[i] for i in self.fpn_scales_from_top]
        rpn_maps_2d = [self.convs_pro2d[i](rpn_maps_3d[i]) for i in range(len(rpn_maps_3d))]
        rpn_maps = rpn_maps_3d + rpn_maps_2d
        rpn_maps = [rpn_maps[i] for i in self.rpn_3d_2d_selector]
        roi_maps = [ups[i] for i in self.roi_scales_from_top]
        for i in range(len(rpn_maps_3d)):
            assert torch.all(rpn_maps_3d[i].spatial_size == torch.tensor(self.rpn_map_sizes[i]))
        return rpn_maps, roi_maps


class GeometryPrefetcher(object):
    """Extension (not in the reference, whose data loader workers only voxelise on the CPU): FPN_Net.prepare for the
    NEXT batch on a helper thread with a stream of its own.  `prepare` reads one site count per grid back from the
    device; on the training thread each of those reads stalls the enqueueing of the forward / backward launches, and
    the main stream runs dry behind it (rocprofv3 timeline of the bench step: 2.9 ms of every 14.5 with nothing queued
    on the main stream while the next batch's geometry was being built).  On the helper thread the reads block only
    the helper: the builders' launches run beside the current batch's forward kernels and the training thread keeps
    enqueueing.  submit(batch) right after net(current batch) returned; wait() before net(next batch).  Everything the
    helper touches is per batch (a new Metadata) or per thread / per stream (SCN's geometry recorder, _hip.workspace)."""

    def __init__(self, net, device=None):
        import queue
        import threading
        self.net = net
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.stream = torch.cuda.Stream(device=self.dev)
        self._q, self._done, self._pending = queue.SimpleQueue(), queue.SimpleQueue(), 0
        self._thread = threading.Thread(target=self._run, name="aabr-geometry-prefetch", daemon=True)
        self._thread.start()

    def _run(self):
        torch.cuda.set_device(self.dev)
        while True:
            item = self._q.get()
            if item is None:
                return
            try:
                with torch.no_grad():
                    self.net.prepare(item, self.stream)
                self._done.put(None)
            except BaseException as e:      # handed to the training thread by wait()
                self._done.put(e)

    def submit(self, batch):
        self._pending += 1
        self._q.put(batch)

    def wait(self):
        """block until every submitted batch has been prepared (its launches are on the helper's stream; forward
        makes the training stream wait for that stream, ioLayers.InputLayer.forward)"""
        while self._pending:
            e = self._done.get()
            self._pending -= 1
            if e is not None:
                raise e

    def close(self):
        self.wait()
        self._q.put(None)
        self._thread.join()
        if getattr(self.net, "_prefetcher", None) is self:
            self.net._prefetcher = None

#!/usr/bin/env python3
"""bench.py -- scenes/s (fwd+bwd) of the sparse-3D hot path on MI355X.

Headline workload = BASELINE.json configs[2] ("walls" config: the default FPN_Net backbone,
maskrcnn_benchmark/config/defaults.py:48-57, SparseConvNet/sparseconvnet/fpn_net.py:13-203), per GPU and step:
  4 synthetic SUNCG-shaped scenes (~80k points each) at 2 cm voxels, C_in = 9
  -> HIP voxel scatter (InputLayer mode 4; hash grid, site numbering and every rule book rebuilt each step)
  -> FPN_Net: 9 scales, 36 SubmanifoldConvolution + 12 Convolution + 8 Deconvolution + 35 BatchNorm(+ReLU)
  -> RPN head (three dense 1x1 layers, torch plumbing as in the reference, rpn_sparse3d.py:81-131; the shared head
     runs once over the rows of all six maps) and a synthetic loss over all anchors
  -> RPN label generation (rpn/loss_3d.py:91-96: 40 ground-truth boxes per scene x the anchors of all six maps,
     criterion-6 IoU + matcher thresholds, one fused kernel on its own stream; the loss's backward waits for it)
  -> per scene: cross-scale top-k(2000) -> fused anchor + BoxCoder3D decode -> rotated-3D NMS (1000)
  -> backward through head and backbone (all weight gradients + the input-feature gradient)
  -> gradient all-reduce (N > 1: RCCL, in buckets started from inside the backward pass) -> SGD update.
The headline runs in fp32, the reference's arithmetic; the same step with bf16 feature storage is reported
beside it (`extras.bf16`).  Inputs are resident in HBM before the timed region.
Pipelining inside a step, all of it real work of that step or the next (nothing cached, nothing skipped):
  * the proposal stage (top-k, decode, NMS: reads forward results only) runs on a side stream underneath the backward
    kernels: its launches go out before the backward pass is enqueued, its one read of counts (a mailbox post,
    _hip.read_back) is taken after the backward pass and the SGD update have been enqueued;
  * the NEXT batch's geometry (voxel grid, strided grids, rule tables, block streams -- coordinates only) is built
    on that side stream from the end of this batch's forward pass on (`FPN_Net.prepare`, the device-side analogue
    of a data-loader prefetch; AABR_BENCH_PREFETCH=0 builds it inline instead); every step's geometry is built from scratch, one
    step ahead; the first step builds its own;
  * the layers between the input layer and the returned maps run through the compiled graph executor
    (sparseconvnet/planExecutor.py: same kernels, arguments and order as the per-layer modules, one launch list
    per pass; AABR_BENCH_COMPILED_GRAPH=0 runs the modules);
  * N > 1: the gradient all-reduce runs in buckets launched from inside the compiled backward pass
    (planExecutor.on_grads_ready) and is waited for before the SGD update;
  * host side: the Python cycle collector is frozen after the warm-up (gc.freeze(): a full collection paused one
    step in a few hundred for 40-65 ms; `timing.gc` reports the pauses).  One process per GPU
(`--gpus N` spawns the ranks itself when not already under torchrun); ranks take different scenes of one global
scene list (weak scaling: per-GPU work fixed); the only collective is the gradient all-reduce.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel instance with the largest share of the step among
the convolution launches of this very workload, found and timed live (HIP events on the launching stream);
`cpu_baseline` is the oracle (CPU port of the reference path) on a bounded sample of the same workload with the
per-stage split of SURVEY 8(d).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec
SCENES_PER_STEP = 4             # bs 4 per GPU (configs[2] / configs[3])
VOXEL_SCALE = 50                # 2 cm
N_POINTS = 80000
N_GT = 40                       # ground-truth walls per scene (label generation)
# cfg.MODEL.RPN.LABEL_AUG_THICKNESS_{Y,Z}_TAR_ANC (config/defaults.py:161-162; rpn/loss_3d.py:351)
LABEL_AUG = {"target_Y": 0.4, "anchor_Y": 0.0, "target_Z": 0.8, "anchor_Z": 0.0}
MIN_TIMED_S = 0.2               # a timed region shorter than this is reported as such (`timed_region_short`)
PMC_PROFILE = "r06_pmc_fetch_write_per_kernel.json"   # committed --pmc passes of this command (tools/tools_pmc.sh)

# RPN constants of the reference config (defaults.py:127-131,159-181)
ANCHOR_SIZES_3D = [[0.4, 1.5, 1.5], [1.5, 1.5, 1.0], [4, 4, 1.5], [0.2, 0.5, 3], [0.4, 1.5, 3], [0.6, 2.5, 3]]
YAWS = (0, -1.57, -0.785, 0.785)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(argv, n, script=None):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (tools/train_net_sparse3d.py:183-190
    is launched by torch.distributed.launch the same way).  This parent never touches the GPU and nothing is
    re-exec'd; children inherit stdout, rank 0 prints the JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("AABR_BENCH_TIMEOUT", "1500"))
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0:
                rc = rc or r
                for q in procs:   # one rank failed: the others would hang in the next collective
                    q.terminate()
        if time.time() > deadline:
            for q in procs:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


SITE_ORDER_DEFAULT = "brick"


# ------------------------------------------------------------------------------------------------ workload
def build_net(scn, torch, dev, dtype, site_order=None):
    torch.manual_seed(0)
    net = scn.FPN_Net([4096, 4096, 512], 3, ["xyz", "color", "normal"], 1,
                      [32, 64, 64, 128, 128, 128, 256, 256, 256], 128, True, [4, 3, 2, 1], [4, 3, 2, 1],
                      [[[2, 2, 2]] * 8, [[2, 2, 2]] * 8],
                      [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]], [1, 2, 3, 4, 5, 6],
                      leakiness=0, voxel_scale=VOXEL_SCALE, bn_momentum=0.95, feature_dtype=dtype).to(dev)
    # every layer between the input layer and the returned maps as one launch list per pass (planExecutor.py):
    # same kernels, arguments and order as the module path, without its per-layer interpreter time
    net.compiled_graph = os.environ.get("AABR_BENCH_COMPILED_GRAPH", "1") != "0"
    # site order of every level (FPN_Net.set_site_order): "brick" = brick-major rows over brick grids (csrc/brick.hip),
    # "first_seen" = the reference's numbering over hash grids; same sites, features and gradients per site either way
    net.set_site_order(site_order or os.environ.get("AABR_BENCH_SITE_ORDER", SITE_ORDER_DEFAULT))

    class RpnHead(torch.nn.Module):
        """SingleConvRPNHead_Sparse3D (rpn_sparse3d.py:81-131): 1x1 conv + ReLU, objectness and box heads --
        dense torch layers in the reference too (nn.Conv2d on [1,C,N,1]); plumbing, not part of the HIP path."""

        def __init__(self, c, a):
            super().__init__()
            self.conv, self.cls_logits, self.bbox_pred = (torch.nn.Linear(c, c), torch.nn.Linear(c, a),
                                                          torch.nn.Linear(c, a * 7))
            for l in (self.conv, self.cls_logits, self.bbox_pred):
                torch.nn.init.normal_(l.weight, std=0.01)
                torch.nn.init.constant_(l.bias, 0)

        def forward(self, f):
            # (round 6: the head's three weight-gradient GEMMs run on 6-29 workgroups, 214 us per step in the trace; cutting
            # their long dimension into a batched GEMM + a sum was measured at 315.4 vs 314.7 scenes/s fp32 -- they overlap
            # the side streams' work -- and costs the host-bound bf16 step 12 more torch launches: not kept)
            t = torch.relu(self.conv(f))
            return self.cls_logits(t).reshape(-1), self.bbox_pred(t).reshape(-1, 7)

    head = RpnHead(128, len(YAWS)).to(dev)
    return net, head


def rpn_constants(torch):
    base = []
    for size in ANCHOR_SIZES_3D:   # generate_anchors_3d_yaws (anchor_generator_sparse3d.py:230-241)
        base.append(torch.tensor([[0.0, 0.0, 0.0] + list(size) + [y] for y in YAWS], dtype=torch.float32))
    # ANCHOR_STRIDE = cumulative SPARSE3D.STRIDE of each map's scale (train_net_sparse3d.py:287-305); rpn maps are
    # the 3-D maps of scales 5,6,7 from the bottom and the z-collapsed maps of scales 4,5,6 (selector [1..6])
    strides = [[2.0 ** s] * 3 for s in (5, 6, 7)] + [[2.0 ** s] * 3 for s in (4, 5, 6)]
    return base, strides


class Workload(object):
    def __init__(self, scn, torch, dp, dev, dtype, rank, world, n_batches, site_order=None):
        import synth_scenes as S
        self.scn, self.torch, self.dev, self.world = scn, torch, dev, world
        self.net, self.head = build_net(scn, torch, dev, dtype, site_order)
        self.flat = dp.FlatParams([self.net, self.head])
        self.flat.broadcast(0)
        self.base, self.strides = rpn_constants(torch)
        # one global scene list, sharded over the ranks (dp.shard_scenes: the sampler the reference lacks)
        n_global = world * n_batches * SCENES_PER_STEP
        mine = dp.shard_scenes(n_global, rank, world, sizes=[N_POINTS] * n_global)
        self.batches = []
        for b in range(n_batches):
            L, F = [], []
            for j, sid in enumerate(mine[b * SCENES_PER_STEP:(b + 1) * SCENES_PER_STEP]):
                l, f = S.make_scene(N_POINTS, 9000 + sid, VOXEL_SCALE)
                import numpy as np
                if os.environ.get("AABR_BENCH_SORT_POINTS") == "1":   # experiment knob (tools/): coherent point order
                    key = (l[:, 0] // 8 * 4096 + l[:, 1] // 8) * 4096 + l[:, 2] // 8
                    o = np.argsort(key, kind="stable")
                    l, f = l[o], f[o]
                L.append(np.concatenate([l, np.full((l.shape[0], 1), j, np.int64)], 1))
                F.append(f)
            import numpy as np
            self.batches.append((torch.as_tensor(np.concatenate(L, 0)).to(dev),
                                 torch.as_tensor(np.concatenate(F, 0)).to(dev).requires_grad_(True)))
        # ground-truth boxes per scene for the label-generation stage (40 synthetic walls each)
        self.targets = []
        for b in range(n_batches):
            self.targets.append([torch.as_tensor(S.make_gt_boxes(N_GT, 7000 + sid)).to(dev)
                                 for sid in mine[b * SCENES_PER_STEP:(b + 1) * SCENES_PER_STEP]])
        self.label_generation = os.environ.get("AABR_BENCH_LABELS", "1") != "0"
        self.grad_buckets = int(os.environ.get("AABR_BENCH_GRAD_BUCKETS", "4"))
        self.last = None
        self.marks = None            # tools/tools_step_timeline.py: [(name, host time, HIP event on the main stream)]
        # AABR_BENCH_SIDE_PRIORITY=1: higher queue priority for the proposal / label streams (chains of small launches
        # beside the backward pass's long workgroups); measured 14.47 -> 14.78 ms, off
        prio = -1 if os.environ.get("AABR_BENCH_SIDE_PRIORITY", "0") != "0" else 0
        self.side = torch.cuda.Stream(device=dev, priority=prio)
        self.lab = torch.cuda.Stream(device=dev, priority=prio)
        self.prefetch_geometry = os.environ.get("AABR_BENCH_PREFETCH", "1") != "0"
        # AABR_BENCH_PREFETCH_THREAD=1: the next batch's geometry on a helper thread (sparseconvnet.GeometryPrefetcher)
        # instead of on this thread between the forward and backward enqueues.  Measured (tools/tools_step_timeline.py,
        # profiles/r03_step_timeline.txt): 14.4 -> 15.5 ms -- the step is bound by the main stream's kernels, this
        # thread already enqueues the forward pass 5 ms ahead of the device, and the helper's interpreter time comes out
        # of this thread's (one GIL); off by default
        self.prefetch_thread = self.prefetch_geometry and os.environ.get("AABR_BENCH_PREFETCH_THREAD", "0") != "0"

    def _mark(self, name):
        if self.marks is not None:
            import time
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, time.perf_counter(), ev))

    def head_loss(self, rpn_maps):
        """The shared RPN head over the rows of all six maps in one call (the reference applies the same head map
        by map, rpn_sparse3d.py:118-131, and concatenates the scales before its loss, :19-77; row-wise layers give
        the same rows either way) and a synthetic loss over all anchors.  Returns the per-map slices the proposal
        stage wants."""
        torch = self.torch
        rows = [m.features.shape[0] for m in rpn_maps]
        o, r = self.head(torch.cat([m.features for m in rpn_maps], 0))
        loss = o.square().mean() + r.square().mean()
        A = o.shape[0] // max(sum(rows), 1)
        return loss, list(o.split([v * A for v in rows])), list(r.split([v * A for v in rows]))

    def forward_backward(self, i, proposals=True, after_backward=None):
        import rpn_glue
        torch = self.torch
        locs, feats = self.batches[i % len(self.batches)]
        ev_start = None
        if self.label_generation:
            ev_start = torch.cuda.Event()
            ev_start.record()          # the main stream is here behind everything the previous step left on the side stream
        threaded = proposals and self.prefetch_thread
        self._mark("step start")
        if threaded:
            self.net.prefetcher().wait()       # this batch's geometry has been built (by the previous step)
        rpn_maps, _ = self.net([locs, feats])
        self._mark("forward enqueued")
        if threaded:
            # the NEXT batch's geometry (voxel grid, strided grids, rule tables, block streams: coordinates only) is built
            # by the helper thread on its own stream while this batch trains -- a data-loader-style prefetch; every step
            # still builds one batch's geometry from scratch, one step ahead
            self.net.prefetcher().submit(self.batches[(i + 1) % len(self.batches)])
        labels = ev_lab = None
        if self.label_generation:
            # RPN label generation (rpn/loss_3d.py:91-96): per scene the criterion-6 IoUs of its ground-truth boxes
            # against the anchors of all six maps + best-match / threshold labels, one library call.  It needs the maps'
            # site lists only (built one step ahead with the rest of the geometry), so it runs on a stream of its own
            # beside the forward pass -- not on the side stream, where it would queue behind the previous step's
            # proposal stage and hold the main stream up (+1.7 ms per step measured); the loss's backward waits for it,
            # as the reference's loss would.
            with torch.no_grad(), torch.cuda.stream(self.lab):
                self.lab.wait_event(ev_start)
                labels = rpn_glue.rpn_label_matches(rpn_maps, self.base, self.strides, float(VOXEL_SCALE),
                                                    self.targets[i % len(self.batches)], LABEL_AUG, 6,
                                                    batch_size=SCENES_PER_STEP)
                ev_lab = torch.cuda.Event()
                ev_lab.record()
        self._mark("labels enqueued")
        loss, objs, regs = self.head_loss(rpn_maps)
        self._mark("head + loss enqueued")
        if ev_lab is not None:
            torch.cuda.current_stream().wait_event(ev_lab)
        # The proposal stage reads only forward results.  It is a chain of small launches (one-workgroup NMS scan,
        # top-k, decode) with a few host reads of counts; enqueued on a side stream AFTER the backward pass has been
        # enqueued on the main one, it runs on otherwise idle CUs underneath the backward kernels and its host
        # reads wait for the side stream only.
        ev_fwd = torch.cuda.Event(enable_timing=os.environ.get("AABR_BENCH_EVFWD_TIMING", "0") == "1")
        ev_fwd.record()
        early = (proposals and self.prefetch_geometry and not threaded
                 and os.environ.get("AABR_BENCH_PREFETCH_EARLY", "1") == "1")
        # brick grids: the prefetch in two halves -- scatter, input bricks and the level pyramid are ENQUEUED here, the
        # counts collected and the rule tables / block streams built after the backward pass has been enqueued, so the
        # host is not waiting while the device builds the pyramid.  Measured (round 5): fp32 step unchanged (device-bound), bf16 537 -> 505 scenes/s -- the tables and block streams then reach the side stream later and the next forward waits for them; off by default (AABR_BENCH_PREPARE_SPLIT=1 turns it on)
        split = early and os.environ.get("AABR_BENCH_PREPARE_SPLIT", "0") == "1"
        if early:
            with torch.no_grad():
                if split:
                    self.net.prepare_begin(self.batches[(i + 1) % len(self.batches)], self.side)
                else:
                    self.net.prepare(self.batches[(i + 1) % len(self.batches)], self.side)
        self._mark("geometry prefetch (inline) done")
        props = finish = None
        launch_first = proposals and os.environ.get("AABR_BENCH_PROPOSALS_FIRST", "1") == "1"

        def launch_proposals(defer):
            with torch.no_grad(), torch.cuda.stream(self.side):
                self.side.wait_event(ev_fwd)
                return rpn_glue.rpn_proposals(rpn_maps, [o.detach() for o in objs], [r.detach() for r in regs],
                                              self.base, self.strides, float(VOXEL_SCALE), 2000, 1000, 0.5, (0.3, 0.3),
                                              batch_size=SCENES_PER_STEP, defer=defer,
                                              batched=os.environ.get("AABR_BENCH_BATCHED_PROPOSALS", "0") != "0")

        if launch_first:
            # every launch of the proposal stage goes out (side stream) BEFORE the backward pass is enqueued; its one
            # read of counts is done after the backward pass and the update have been enqueued
            finish = launch_proposals(True)
            self._mark("proposal launches out")
        loss.backward()
        self._mark("backward enqueued")
        feats.grad = None
        if after_backward is not None:
            after_backward()      # N = 1: the update; N > 1: the gradient all-reduce starts here and runs under the proposal stage
        if split:
            with torch.no_grad():
                self.net.prepare_end()
            self._mark("geometry prefetch, second half done")
        if proposals:
            main = torch.cuda.current_stream()
            if self.prefetch_geometry and not early and not threaded:
                # the NEXT batch's geometry (voxel grid, strided grids, rule tables, block streams: coordinates only)
                # is built on the side stream while this batch's backward runs -- a data-loader-style prefetch; every
                # step still builds its geometry from scratch, one step ahead
                with torch.no_grad():
                    self.net.prepare(self.batches[(i + 1) % len(self.batches)], self.side)
            if launch_first:
                with torch.no_grad(), torch.cuda.stream(self.side):
                    props = finish() if callable(finish) else finish
            else:
                props = launch_proposals(False)
            self._mark("proposals stage left")
            if os.environ.get("AABR_BENCH_JOIN_SIDE", "1") == "1":
                main.wait_stream(self.side)
            self._mark("main waits for the proposal stream")
        self.last = (rpn_maps, props, labels)
        return loss

    def _update_now(self):
        self.flat.sgd_step(1e-5, 1)
        self._mark("update enqueued")

    def step(self, i):
        self._mark("step() entered")
        self.flat.zero_grad()
        if self.world > 1 and self.grad_buckets > 1:
            # bucketed gradient all-reduce (RCCL over xGMI): the compiled backward hands its gradient buffer over in
            # `grad_buckets` slices as they become final; each slice's all-reduce starts at once, in place, and runs
            # under the rest of the backward pass and the proposal stage; the update waits for all of them
            from sparseconvnet import planExecutor
            planExecutor.grad_segments = self.grad_buckets
            self.flat.begin_bucketed()
            try:
                self.forward_backward(i)
            except BaseException:
                self.flat.abort_bucketed()     # never leave the hook armed behind a failed backward
                raise
            self.flat.finish_bucketed(1e-5, self.world)
        elif self.world > 1:
            # one flat all-reduce, launched asynchronously right after backward; the proposal stage (top-k, decode, NMS:
            # no dependence on the gradients) runs while it is in flight; the update waits for it
            self.forward_backward(i, after_backward=self.flat.start_allreduce)
            self.flat.finish_update(1e-5, self.world)
        elif os.environ.get("AABR_BENCH_UPDATE_LAST", "0") == "1":      # round-2 order, for the A/B
            self.forward_backward(i)
            self.flat.sgd_step(1e-5, 1)
            self._mark("update enqueued")
        else:
            # the update is enqueued right behind the backward pass, BEFORE this thread goes to read the proposal
            # stage's counts: the main stream runs forward, backward, update back to back instead of idling between the
            # last backward kernel and an update the host had not enqueued yet (1.1 ms of 14.4 per step measured,
            # profiles/r03_step_timeline.txt).  The proposal stage reads forward activations only.
            self.forward_backward(i, after_backward=self._update_now)


# ------------------------------------------------------------------------------------------------ measurement
def hip_time(torch, fn, group, reps):
    """average duration of fn() from HIP events on the launching (current) stream; events bracket GROUPS of
    back-to-back launches (an event pair around one ~20 us launch reads 2-3 us high)"""
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        for _ in range(group):
            fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) / group for a, b in evs)
    ms = ms[: max(1, len(ms) * 3 // 4)]
    return sum(ms) / len(ms) * 1e-3


def device_time(torch, fn, group=4, reps=6, block_cycles=6000000):
    """device-side duration of fn()'s launches with the HOST taken out: a busy-wait kernel holds the stream
    (torch.cuda._sleep, ~2-3 ms) while `group` calls are enqueued behind it, HIP events bracket the calls -- they run
    back to back once the wait ends, so the elapsed time has no enqueue gaps in it (hip_time with group 1 measures a
    short chain of launches at the pace the interpreter issues them).  Returns seconds per call."""
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(block_cycles)
        a.record()
        for _ in range(group):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / group)
    out.sort()
    out = out[: max(1, len(out) * 3 // 4)]
    return sum(out) / len(out) * 1e-3


def conv_kernel_table(torch, wl, dtype, max_rows=None):
    """Every convolution launch of one training step (forward, input-gradient, weight-gradient), grouped by
    (kernel kind, planes, rule book); each distinct instance re-launched alone and timed with HIP events.
    Returns rows sorted by their share of the step."""
    import _hip
    from _hip import ptr, stream, check
    from sparseconvnet import SCN
    lib = _hip.load()
    SCN.trace = []
    wl.flat.zero_grad()
    compiled, wl.net.compiled_graph = wl.net.compiled_graph, False   # the module path reports its launches
    wl.forward_backward(0, proposals=False)
    wl.net.compiled_graph = compiled
    torch.cuda.synchronize()
    tr, SCN.trace = SCN.trace, None
    groups = {}
    for kind, n_in, n_out, gather, rows_in, flags, dt in tr:
        key = (kind, n_in, n_out, id(gather), flags, dt)
        g = groups.setdefault(key, dict(kind=kind, n_in=n_in, n_out=n_out, gather=gather, rows_in=rows_in,
                                        flags=flags, dtype=dt, calls=0))
        g["calls"] += 1
    rows = []
    dev = wl.dev
    for g in groups.values():
        ga, n_in, n_out, rows_in = g["gather"], g["n_in"], g["n_out"], g["rows_in"]
        if max_rows is not None and ga.rows > max_rows:      # dev tools: the coarse scales only
            continue
        R = float(sum(ga.rule_counts()))
        bf = g["dtype"] == torch.bfloat16
        fdt = torch.bfloat16 if bf else torch.float32
        if g["kind"] == "fwd":
            inp = torch.randn((rows_in, n_in), device=dev).to(fdt)
            out = torch.empty((ga.rows, n_out), device=dev, dtype=fdt)
            tr_ = g["flags"] & 1
            w = torch.randn((ga.vol, 1, n_out, n_in) if tr_ else (ga.vol, 1, n_in, n_out), device=dev) * 0.05
            if bf:
                wpack = torch.empty(lib.aabr_conv_wpack_bf16_elems(ga.vol, w.size(2), w.size(3)), device=dev,
                                    dtype=torch.bfloat16)
                conv = lib.aabr_conv_forward_bf16
            else:
                wpack = torch.empty(lib.aabr_conv_wpack_floats(ga.vol, w.size(2), w.size(3)), device=dev)
                conv = lib.aabr_conv_forward
            narrow = SCN.narrow_ok(n_in, n_out, rows_in, ga.rows, ga.vol, bf)        # the dispatch of SCN._conv_fwd
            tile_rows = 0 if narrow else SCN.wide_tile_rows(n_in, n_out, rows_in, ga.rows, ga.vol, bf)
            if narrow:
                nfn = lib.aabr_conv_forward_narrow_bf16 if bf else lib.aabr_conv_forward_narrow

                def fn():
                    check(nfn(ptr(inp), rows_in, ptr(out), ga.rows, ptr(ga.table), ga.vol, ptr(w), None, g["flags"] & 3,
                              stream()))
            elif tile_rows and bf:
                ncb_ = 2 if (n_out % 128 == 0 and n_in <= 128) else 1      # 128-column slabs (conv_wide.hip)
                g["grid_threads"] = ((ga.rows + tile_rows - 1) // tile_rows) * (n_out // (64 * ncb_)) * 256
                blocks = ga.blocks_wide(tile_rows)
                wt = torch.empty_like(wpack)
                if tr_:     # w is the layer's own weight; the input-gradient launch reads its transposed pack
                    check(lib.aabr_conv_pack_weights2_bf16(ptr(w), ga.vol, w.size(2), w.size(3), ptr(wt), ptr(wpack),
                                                           stream()))
                else:
                    check(lib.aabr_conv_pack_weights2_bf16(ptr(w), ga.vol, w.size(2), w.size(3), ptr(wpack), ptr(wt),
                                                           stream()))

                def fn():
                    check(lib.aabr_conv_forward_wide_bf16(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows,
                                                          ptr(blocks), tile_rows, ga.vol, None, g["flags"] & 3,
                                                          ptr(wpack), stream()))
            elif tile_rows:
                g["grid_threads"] = ((ga.rows + tile_rows - 1) // tile_rows) * (n_out // 64) * 256
                blocks = ga.blocks_wide(tile_rows)
                check(lib.aabr_conv_pack_weights(ptr(w), ga.vol, n_in, n_out, tr_, ptr(wpack), stream()))

                def fn():
                    check(lib.aabr_conv_forward_wide(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows, ptr(blocks),
                                                     tile_rows, ga.vol, None, g["flags"] & 3, ptr(wpack), stream()))
            elif SCN.wide_split(n_in, n_out, rows_in, ga.rows, ga.vol, bf):
                Ts, P = SCN.wide_split(n_in, n_out, rows_in, ga.rows, ga.vol, bf)
                blocks = ga.blocks_wide(Ts)
                scratch = torch.empty(P * ga.rows * n_out, device=dev)
                if bf:
                    wt = torch.empty_like(wpack)
                    a_, b_ = (wt, wpack) if tr_ else (wpack, wt)   # the input-gradient launch reads the transposed pack
                    check(lib.aabr_conv_pack_weights2_bf16(ptr(w), ga.vol, w.size(2), w.size(3), ptr(a_), ptr(b_), stream()))

                    def fn():
                        check(lib.aabr_conv_forward_wide_split_bf16(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows,
                                                                    ptr(blocks), Ts, ga.vol, None, g["flags"] & 3,
                                                                    ptr(wpack), P, ptr(scratch), stream()))
                else:
                    check(lib.aabr_conv_pack_weights(ptr(w), ga.vol, n_in, n_out, tr_, ptr(wpack), stream()))

                    def fn():
                        check(lib.aabr_conv_forward_wide_split(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows,
                                                               ptr(blocks), Ts, ga.vol, None, g["flags"] & 3, ptr(wpack),
                                                               None, P, ptr(scratch), stream()))
            else:
                blocks = ga.blocks()
                check(conv(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows, ptr(blocks), ga.vol, ptr(w), None,
                           g["flags"], ptr(wpack), stream()))   # packs the weights once

                def fn():
                    check(conv(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows, ptr(blocks), ga.vol, ptr(w), None,
                               g["flags"] | 4, ptr(wpack), stream()))
        else:
            inp = torch.randn((rows_in, n_in), device=dev).to(fdt)
            d_out = torch.randn((ga.rows, n_out), device=dev).to(fdt)
            dW = torch.empty((ga.vol, n_in, n_out), device=dev)
            pairs = ga.pairs()
            mc = ga.max_chunks(n_in, n_out)
            scratch = torch.empty(max(1, lib.aabr_conv_dw_scratch_floats(mc, n_in, n_out)), device=dev)
            fnc = lib.aabr_conv_backward_weight_bf16 if bf else lib.aabr_conv_backward_weight

            def fn():
                check(fnc(ptr(inp), n_in, ptr(d_out), n_out, ga.rows, ptr(pairs), ga.vol, mc, ptr(dW), None,
                          ptr(scratch), stream()))
        fn()
        variant = lib.aabr_conv_last_variant().decode()
        sec = hip_time(torch, fn, 4, 6)
        flops = 2.0 * R * n_in * n_out
        rows.append(dict(kind=g["kind"], kernel=variant, n_in=n_in, n_out=n_out, vol=ga.vol, rows_out=int(ga.rows),
                         rules=int(R), calls_per_step=g["calls"], launch_us=round(sec * 1e6, 2),
                         step_us=round(sec * 1e6 * g["calls"], 1), tflops=round(flops / sec / 1e12, 2),
                         flops_per_launch=flops, grid_threads=g.get("grid_threads", 0)))
        del inp
    rows.sort(key=lambda r: -r["step_us"])
    return rows


def pmc_traffic(kernel_name, grid_threads):
    """HBM bytes per launch of the (kernel, grid size) instance from the committed PMC passes of THIS command
    (profiles/, separate --pmc runs, tools/tools_pmc.sh; 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 note of
    MI355X_MICROARCH.md).  Returns (bytes, source) or (None, reason) -- never a number for a different kernel."""
    name = PMC_PROFILE
    path = os.path.join(REPO, "profiles", name)
    try:
        pm = json.load(open(path))["kernels"]
    except Exception:
        return None, "no committed PMC profile for this round"
    key = "%s|grid=%d" % (kernel_name.replace(" ", ""), grid_threads)
    v = pm.get(key)
    if v is None:
        # the profiler prints every template argument (k_conv_cs<4,0,1,false,1,2,false>), aabr_conv_last_variant() the
        # leading ones that tell the instances apart (k_conv_cs<4,0,1>): same instance when the leading arguments, the
        # storage type and the grid size all match
        kn = kernel_name.replace(" ", "")
        stem, bf16 = kn.split(",bf16")[0].rstrip(">"), ",bf16" in kn
        x128 = ",x128" in kn
        cands = [k for k in pm if k.endswith("|grid=%d" % grid_threads) and k.startswith(stem + ",")
                 and (",true," in k.split("|")[0]) == bf16]
        if bf16:
            cands = [k for k in cands if (k.split("|")[0].split(",")[4] == "2") == x128]
        if len(cands) == 1:
            key, v = cands[0], pm[cands[0]]
    if v is None:
        return None, "instance %s not in the committed PMC profile (dispatch or workload changed since it was taken)" % key
    return int((2.0 * v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024), \
        "committed profile profiles/%s, entry %s (%d launches)" % (name, key, v["launches"])


def pmc_kernels_traffic(prefixes, source=None):
    """sum over the committed PMC profile's entries whose kernel name starts with one of `prefixes` of the bytes moved
    per launch (2 x FETCH + WRITE), largest grid of each kernel -- the voxel-scatter stage's HBM traffic; None when
    the profile is missing.  `source`: only entries of that command (None = the default command's own passes; the profile
    also carries the instances of `--dtype bf16` and `--config 4`, marked `from`)"""
    try:
        pm = json.load(open(os.path.join(REPO, "profiles", PMC_PROFILE)))["kernels"]
    except Exception:
        return None, "no committed PMC profile for this round"
    best = {}
    for k, v in pm.items():
        nm, _, g = k.partition("|grid=")
        if not any(nm.startswith(p) for p in prefixes) or "backward" in nm or v.get("from") != source:
            continue
        if nm not in best or int(g) > best[nm][0]:
            best[nm] = (int(g), (2.0 * v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024)
    if not best:
        return None, "no scatter kernels in profiles/%s" % PMC_PROFILE
    return int(sum(b for _, b in best.values())), "profiles/%s: %s" % (PMC_PROFILE, ", ".join(sorted(best)))


def scatter_block(torch, scn, locs, feats, tag):
    """voxel scatter (A1+A2) on one point list: device time of the geometry half (hash grid + first-seen numbering +
    chains: csrc/voxel_scatter.hip) and of the feature half (ordered mean), host taken out (device_time); algorithmic
    bytes N*(32 + 4*C_in) + V*(4*C_in + 16)"""
    import _hip
    from _hip import ptr, stream, check
    from sparseconvnet import SCN
    lib = _hip.load()
    dev = locs.device
    SP = torch.LongTensor([4096, 4096, 512])
    fdet = feats.detach()
    keep = []

    order = os.environ.get("AABR_BENCH_SITE_ORDER", SITE_ORDER_DEFAULT)

    def sites():
        md = SCN.Metadata_3()
        md.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
        keep.append(md)
        del keep[:-6]
    with torch.no_grad():
        t_sites = device_time(torch, sites)
        t_brick = 0.0
        if order == "brick" and SCN.brick_scatter:
            # brick-major rows: the scatter goes straight into the brick grid (csrc/brick.hip): points converted + extent,
            # input level built from the points, every point handed its row / first point / chain -- no hash table
            md0 = SCN.Metadata_3("brick")
            md0.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
            ext = md0._pending["meta"].tolist()[8:12]

            def sites():
                md = SCN.Metadata_3("brick")
                md.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
                p_ = md._pending
                keep.append((md, md._brick_scatter_launch(p_["piece"], md.input["n"], md.input["spatial"], ext, dev)))
                del keep[:-6]
            t_sites = device_time(torch, sites)
        elif order == "brick":
            # (AABR_BRICK_SCATTER=0) the hash scatter followed by the brick level + renumbering of its sites
            mdf = SCN.Metadata_3()
            mdf.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
            pend = mdf._pending
            m = pend["meta"].tolist()

            def brickify():
                md2 = SCN.Metadata_3("brick")
                md2.input = dict(mdf.input)
                keep.append((md2._brickify_input(pend, m[0], m[8:12]), md2))
                del keep[:-6]
            t_brick = device_time(torch, brickify)
            t_sites += t_brick
        md = SCN.Metadata_3(order)
        V = md.inputLayer(SP, locs, 4, 4, dev)
        il = md.input
        out = torch.empty((V, fdet.shape[1]), device=dev)

        def mean():
            check(lib.aabr_input_layer_forward(ptr(fdet), ptr(out), V, fdet.shape[1], ptr(il["first_pt"]),
                                               ptr(il["cnt_extra"]), ptr(il["head"]), ptr(il["nxt"]), ptr(il["last_pt"]),
                                               4, ptr(il["meta"]), stream()))
        t_mean = device_time(torch, mean)
        inp = scn.InputLayer(3, [4096, 4096, 512], mode=4)
        inp.site_order = order
        t_call = hip_time(torch, lambda: inp([locs, fdet]), 1, 10)
    n, c = int(locs.shape[0]), int(fdet.shape[1])
    by = n * (32 + 4 * c) + V * (4 * c + 16)
    t = t_sites + t_mean
    return dict(workload=tag, points=n, sites=int(V), bytes=by, insert_form=("brick-native (no hash table)" if (order == "brick" and SCN.brick_scatter) else int(SCN.scatter_variant)), site_order=order,
                device_seconds=round(t, 7), sites_seconds=round(t_sites, 7), brick_seconds=round(t_brick, 7),
                mean_seconds=round(t_mean, 7),
                device_gbs=round(by / t / 1e9, 2), device_frac_of_hbm_peak=round(by / t / 1e9 / PEAK_HBM_GBS, 5),
                seconds=round(t_call, 7), achieved_gbs=round(by / t_call / 1e9, 2),
                frac_of_hbm_peak=round(by / t_call / 1e9 / PEAK_HBM_GBS, 5))


def stage_rooflines(torch, scn, wl, table, V0):
    """SURVEY 8(d)'s per-stage figures beside the headline: rule-book build (bytes + probes/s), convolution forward /
    weight gradient (flop-weighted over the step's launches), BatchNorm forward / backward (HBM fraction), rotated NMS
    (pairs/s).  Each stage is timed alone on this workload's own operands, device side (device_time)."""
    import _hip
    import _nms
    import synth_scenes as S
    from _hip import ptr, stream, check
    from sparseconvnet import SCN
    lib = _hip.load()
    dev = wl.dev
    out = {}
    locs, feats = wl.batches[0]
    with torch.no_grad():
        # --- submanifold rule table of the input grid, k = 3: 27 probes per site into the hash grid
        order = wl.net.site_order
        md = SCN.Metadata_3(order)
        V = md.inputLayer(torch.LongTensor([4096, 4096, 512]), locs, 4, 4, dev)
        g = md.grids[(4096, 4096, 512)]
        table_ = torch.empty((27, g.V), dtype=torch.int32, device=dev)
        counts = torch.empty(27 * ((g.V + 255) // 256), dtype=torch.int32, device=dev)
        fs = _hip.i32x3((3, 3, 3))

        def subm():
            if g.brick is not None:
                check(lib.aabr_brick_submanifold_table(ptr(g.coords), g.V, g.brick.dims_c(), g.brick.dir_ptr(),
                                                       g.brick.bricks_ptr(), fs, ptr(table_), ptr(counts), stream()))
            else:
                check(lib.aabr_submanifold_table(ptr(g.coords), g.V, ptr(g.keys), g.cap, fs, ptr(table_), ptr(counts),
                                                 stream()))
        t = device_time(torch, subm)
        R = int((table_ >= 0).sum().item())
        by = 16 * g.V + 4 * 27 * g.V
        out["rulebook_build"] = dict(what="%s, k = 3, input grid" % ("aabr_brick_submanifold_table (brick grid: two dependent "
                                                                     "16-byte loads per look-up)" if g.brick is not None
                                                                     else "aabr_submanifold_table (hash grid)"),
                                     site_order=order, sites=int(g.V), rules=R,
                                     seconds=round(t, 7), bytes=by, gbs=round(by / t / 1e9, 1),
                                     frac_of_hbm_peak=round(by / t / 1e9 / PEAK_HBM_GBS, 4),
                                     probes_per_s=round(27 * g.V / t, 1),
                                     note="bytes = 16 V (site list) + 4 vol V (gather table written); SURVEY 8d's 2R*4 "
                                          "pair form is produced from the table by k_fill_offset_pairs")
        # --- BatchNorm forward / backward on the largest 32- and 128-plane maps of the step
        bn = {}
        for rows, planes in ((V0, 32), (max((r["rows_out"] for r in table if r["n_out"] == 128), default=V0 // 4), 128)):
            x = torch.randn((rows, planes), device=dev)
            y = torch.empty_like(x)
            sm, si = torch.empty(planes, device=dev), torch.empty(planes, device=dev)
            rm, rv = torch.zeros(planes, device=dev), torch.ones(planes, device=dev)
            w, b = torch.ones(planes, device=dev), torch.zeros(planes, device=dev)
            t_f = device_time(torch, lambda: SCN.BatchNormalization_updateOutput(x, y, sm, si, rm, rv, w, b, 1e-4, 0.95,
                                                                               True, 0.0))
            dy, dx = torch.randn_like(x), torch.empty_like(x)
            dw, db = torch.empty(planes, device=dev), torch.empty(planes, device=dev)
            t_b = device_time(torch, lambda: SCN.BatchNormalization_backward(x, dx, y, dy, sm, si, rm, rv, w, b, dw, db,
                                                                            0.0))
            bf, bb = 3 * 4 * rows * planes, 5 * 4 * rows * planes
            bn["%dx%d" % (rows, planes)] = dict(fwd_us=round(t_f * 1e6, 1), fwd_gbs=round(bf / t_f / 1e9, 1),
                                                fwd_frac_of_hbm_peak=round(bf / t_f / 1e9 / PEAK_HBM_GBS, 4),
                                                bwd_us=round(t_b * 1e6, 1), bwd_gbs=round(bb / t_b / 1e9, 1),
                                                bwd_frac_of_hbm_peak=round(bb / t_b / 1e9 / PEAK_HBM_GBS, 4))
        out["batchnorm"] = dict(bn, note="algorithmic bytes: forward 3 * 4 V C, backward 5 * 4 V C (SURVEY 8d); unfused "
                                         "entry points, timed alone")
        # --- rotated NMS of one 2000-box proposal set
        b7, sc = S.make_nms_boxes(2000, 0)
        o = (-sc).argsort()
        bd = torch.as_tensor(b7[o]).to(dev)
        t_n = device_time(torch, lambda: _nms.rotate_nms_sorted(bd, 0.5, 1000, True, lazy=True), group=2)
        out["rotated_nms"] = dict(boxes=2000, seconds=round(t_n, 7), pairs=2000 * 1999 // 2,
                                  pairs_per_s=round(2000 * 1999 / 2 / t_n, 1),
                                  note="mask (A15 IoU pre-filter + exact clip decision) + greedy scan, device side; "
                                       "VALU / latency bound, no roofline fraction claimed (SURVEY 8d)")
    for kind, name in (("fwd", "conv_forward_type"), ("dw", "conv_weight_gradient")):
        fl = sum(r["flops_per_launch"] * r["calls_per_step"] for r in table if r["kind"] == kind)
        us = sum(r["step_us"] for r in table if r["kind"] == kind)
        if us:
            bf = any("bf16" in r["kernel"] for r in table if r["kind"] == kind)
            peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
            out[name] = dict(flop_weighted_tflops=round(fl / us / 1e6, 2), frac_of_mfma_peak=round(fl / us / 1e6 / peak, 4),
                             step_us=round(us, 1), launches=sum(r["calls_per_step"] for r in table if r["kind"] == kind))
    return out


def roofline_block(torch, wl, dtype, step_us):
    """`roofline` object for the workload's storage dtype: the convolution instance with the largest share of the
    step, timed alone with HIP events, priced against the MFMA peak of the arithmetic it issues."""
    table = conv_kernel_table(torch, wl, dtype)
    top = table[0]
    bf = "bf16" in top["kernel"]
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
    traffic, src = pmc_traffic(top["kernel"], top["grid_threads"])
    rf = dict(
        kernel="%s (%s %d->%d, vol %d, %d output rows, %d rules; %d launches per step = %.0f us of the %.0f us step)"
               % (top["kernel"], top["kind"], top["n_in"], top["n_out"], top["vol"], top["rows_out"], top["rules"],
                  top["calls_per_step"], top["step_us"], step_us),
        bound="mfma", achieved=top["tflops"], peak=peak, unit="TFLOP/s", frac=round(top["tflops"] / peak, 5),
        traffic=traffic, traffic_source=src, launch_us=top["launch_us"],
        algorithmic_flops_per_launch=top["flops_per_launch"],
        kernel_source="aabr_conv_last_variant() of the timed launch")
    return rf, table


def cpu_baseline(wl, torch, budget_s=25.0):
    """oracle (CPU port of the reference path; contraction / BN / input kernels pinned to the reference's own
    compiled CPU kernels, tests/test_oracle_ref_kernels.py) on a bounded sample of the same workload, with the
    per-stage split of SURVEY 8(d)"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np
    import oracle_lib as O
    import ref_net
    import synth_scenes as S
    if _ORIG_AFFINITY:
        try:
            os.sched_setaffinity(0, _ORIG_AFFINITY)
        except Exception:
            pass
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    O.set_threads(max(1, min(ncpu, int(os.environ.get("AABR_CPU_THREADS", "16")))))
    P = ref_net.fpn_params(wl.net)
    done, t0, stages = 0, time.perf_counter(), {}
    n_cpu = min(N_POINTS, 80000)      # --config 4: the sample stays S80k scenes of the same network (bounded CPU time)
    while True:
        l, f = S.make_scene(n_cpu, 9000 + done, VOXEL_SCALE)
        locs = np.concatenate([l, np.zeros((l.shape[0], 1), np.int64)], 1)
        t_s = time.perf_counter()
        fo = ref_net.FpnOracle(P, (4096, 4096, 512), [[2, 2, 2]] * 8, [[2, 2, 2]] * 8,
                               [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]])
        rpn, _ = fo.forward(locs, f)
        t_f = time.perf_counter()
        fo.backward([2.0 * m.v / m.v.size for m in rpn])
        t_b = time.perf_counter()
        for k, v in fo.timing.items():
            stages[k] = stages.get(k, 0.0) + v
        stages["backbone_fwd"] = stages.get("backbone_fwd", 0.0) + (t_f - t_s)
        stages["backbone_fwd_bwd"] = stages.get("backbone_fwd_bwd", 0.0) + (t_b - t_s)
        b7, sc = S.make_nms_boxes(2000, done)
        t_n = time.perf_counter()
        O.rotate_nms_3d(b7, sc, 2000, 1000, 0.5)
        stages["iou_nms_2000"] = stages.get("iou_nms_2000", 0.0) + time.perf_counter() - t_n
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 8:
            break
    total = stages["backbone_fwd_bwd"] + stages["iou_nms_2000"]
    return dict(value=round(done / total, 4), unit="scenes/s", cores=O.num_threads(), kind="port",
                sample="%d x S80k@2cm scene through the whole FPN_Net fwd+bwd + one 2000-box rotated NMS, %.1f s of CPU "
                       "work, OpenMP over %d threads (scene generation excluded)" % (done, total, O.num_threads()),
                stage_seconds_per_scene={k: round(v / done, 4) for k, v in sorted(stages.items())})


_ORIG_AFFINITY = None


def pin_host_threads(torch, local_rank, ncores=4):
    """Bind this process (launch thread + autograd thread) to a few cores of the GPU's NUMA node: unbound, the
    launch thread wanders over a 2-socket host and the host-bound part of a step moves by 10-30 % run to run.
    Best effort: any failure leaves the affinity untouched."""
    global _ORIG_AFFINITY
    try:
        allowed = sorted(os.sched_getaffinity(0))
        _ORIG_AFFINITY = set(allowed)
        if len(allowed) <= ncores:
            return None
        cand = allowed
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            bdf = None
        if bdf:
            node_file = "/sys/bus/pci/devices/%s/numa_node" % str(bdf).lower()
            if os.path.exists(node_file):
                node = int(open(node_file).read().strip())
                if node >= 0:
                    cl = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
                    cpus = []
                    for part in cl.split(","):
                        a, _, b = part.partition("-")
                        cpus += list(range(int(a), int(b or a) + 1))
                    cand = [c for c in cpus if c in set(allowed)] or allowed
        nblocks = max(1, len(cand) // ncores)
        b0 = (local_rank % nblocks) * ncores
        cores = set(cand[b0:b0 + ncores])
        os.sched_setaffinity(0, cores)
        return sorted(cores)
    except Exception:
        return None


def _pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def settle(torch, dist, wl, i0, world, dev, max_s=6.0, max_steps=400, group=5, tol=0.03):
    """DISCLOSED pre-warm (reported as `prewarm_steps` / `prewarm_s` in the line): untimed full training steps, in
    groups of `group`, until two consecutive groups agree within `tol` -- a fresh process starts with the GPU at
    its idle clock, a cold caching allocator and cold host caches, and a 14 ms step measured 0.1 s after process
    start is 30-40 % slow (round 2: 19.8 ms driver-timed against 14.2 in steady state).  Bounded by `max_s`
    seconds / `max_steps` steps.  With several ranks the group time and the elapsed time are MAX-reduced so that
    every rank runs the same number of steps (each step holds a collective).
    Returns (steps run, seconds, ms per step of the last group, per-group log)."""
    t_start = time.perf_counter()
    prev, n, log = None, 0, []
    while n < max_steps:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(group):
            wl.step(i0 + n + k)
        torch.cuda.synchronize()
        cur, spent = (time.perf_counter() - t0) / group * 1e3, time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([cur, spent], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cur, spent = float(t[0].item()), float(t[1].item())
        n += group
        log.append(round(cur, 3))
        done = prev is not None and abs(cur - prev) <= tol * min(cur, prev)
        prev = cur
        if done or spent >= max_s:
            break
    return n, time.perf_counter() - t_start, prev, log


def _region(torch, dist, wl, n, i0, world, dev):
    """n steps between barrier + synchronize on both sides; per-step HIP events and host stamps, no sync inside"""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = [0.0] * (n + 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    host[0] = t0
    for i in range(n):
        wl.step(i0 + i)
        evs[i + 1].record()
        host[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    host_ms = [(host[i + 1] - host[i]) * 1e3 for i in range(n)]
    return el, dev_ms, host_ms


def timed_steps(torch, dist, wl, steps, warmup, world, dev, min_timed_s=0.0, prewarm=True):
    """`warmup` untimed steps, the disclosed settle loop, then EXACTLY `steps` timed steps between barrier +
    synchronize on both sides, MAX over ranks (the contract's region: `value` comes from it).  Every timed step also
    gets a host timestamp and a HIP event at its end on the launching stream (no synchronisation inside the
    region), so the line shows the distribution of step times and the first five.  When that region is shorter
    than `min_timed_s`, a SECOND region long enough to fill it is timed the same way right after and reported
    beside the first (`timing.extended`): a cross-check of a 0.3 s headline against >= 1 s of the same steps."""
    import gc
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize()
    info = {"prewarm_steps": 0, "prewarm_s": 0.0}
    # Python's cycle collector: a full (generation-2) collection walks every tracked object of the process -- the
    # network, the templates, torch's own module state: 60-70 ms measured, once every few hundred steps, i.e. one step
    # of 72 ms among steps of 8.3 (profiles/r03_bench_line_bf16.json, `extended.step_ms_max`).  The objects that exist
    # after the warm-up live as long as the process: gc.freeze() moves them to the permanent generation, so the
    # collections that do run inside the loop look at the step's own garbage only.  AABR_BENCH_GC=default leaves the
    # collector as it is; the pauses are counted either way (`timing.gc`).
    gc_mode = os.environ.get("AABR_BENCH_GC", "freeze")
    gc_log = []
    t_gc = [0.0]

    def _gc_cb(phase, inf):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            gc_log.append((inf.get("generation", -1), (time.perf_counter() - t_gc[0]) * 1e3))

    if gc_mode == "freeze":
        gc.collect()
        gc.freeze()
    gc.callbacks.append(_gc_cb)
    i0 = warmup
    if prewarm:
        n, sec, est, log = settle(torch, dist, wl, i0, world, dev)
        i0 += n
        info.update(prewarm_steps=n, prewarm_s=round(sec, 3),
                    prewarm_group_ms=log if len(log) <= 10 else log[:6] + ["..."] + log[-3:])
    el_local, dev_ms, host_ms = _region(torch, dist, wl, steps, i0, world, dev)
    i0 += steps
    el = el_local
    if world > 1:
        t = torch.tensor([el, -el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0].item())
        info["rank_ms_per_step_max"] = round(el / steps * 1e3, 3)
        info["rank_ms_per_step_min"] = round(-float(t[1].item()) / steps * 1e3, 3)
    info["step_ms"] = dict(p50=round(_pct(dev_ms, 0.5), 3), p90=round(_pct(dev_ms, 0.9), 3), min=round(min(dev_ms), 3),
                           max=round(max(dev_ms), 3), first5=[round(v, 3) for v in dev_ms[:5]],
                           host_enqueue_p50=round(_pct(host_ms, 0.5), 3),
                           note="device-side period between consecutive end-of-step HIP events (rank 0); "
                                "host_enqueue = host time per step (includes its blocking reads)")
    info["rank0_s"] = round(el_local, 4)
    if min_timed_s > 0 and el < min_timed_s:
        n2 = int(min_timed_s / (el / steps) + 0.999)
        if world > 1:   # every rank runs the same number of steps (each holds a collective)
            t = torch.tensor([n2], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n2 = int(t.item())
        el2, dev2, _ = _region(torch, dist, wl, n2, i0, world, dev)
        if world > 1:
            t = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        info["extended"] = dict(steps=n2, timed_region_s=round(el2, 3), ms_per_step=round(el2 / n2 * 1e3, 3),
                                scenes_per_s=round(world * n2 * SCENES_PER_STEP / el2, 2),
                                step_ms_p50=round(_pct(dev2, 0.5), 3), step_ms_max=round(max(dev2), 3),
                                ratio_to_headline=round((el2 / n2) / (el / steps), 4))
    gc.callbacks.remove(_gc_cb)
    info["gc"] = dict(mode=gc_mode, collections=len(gc_log), full_collections=sum(1 for g, _ in gc_log if g == 2),
                      pause_ms_max=round(max([m for _, m in gc_log] or [0.0]), 2),
                      pause_ms_total=round(sum(m for _, m in gc_log), 2),
                      note="Python cycle-collector runs between the end of the warm-up and the end of the last timed "
                           "region (pre-warm included)")
    log_path = os.environ.get("AABR_BENCH_STEP_LOG")
    if log_path:
        with open(log_path, "w") as f:
            json.dump(dict(dev_ms=dev_ms, host_ms=host_ms, info=info), f)
    return el, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct resident batches per rank, cycled (8: the timed region does not keep seeing the same "
                         "two shapes)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="BASELINE.json configs index: 2 (default, the headline) = 4 x S80k @ 2 cm per GPU and step; "
                         "4 = one 1.5 M-point scene @ 2 cm per step (the rule-book stress; bf16 storage unless --dtype "
                         "f32), same network and step")
    ap.add_argument("--min-timed-s", type=float, default=1.0,
                    help="when the --steps region is shorter than this, a second region of at least this many "
                         "seconds is timed right after it and reported as `timing.extended` (0 = off)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the disclosed settle loop before the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="feature storage of the HEADLINE loop (default: f32 for --config 2, bf16 for --config 4)")
    args = ap.parse_args()
    global SCENES_PER_STEP, N_POINTS
    if args.config == 4:
        SCENES_PER_STEP, N_POINTS = 1, 1500000
        if args.batches > 2:
            args.batches = 2
    if args.dtype is None:
        args.dtype = "bf16" if args.config == 4 else "f32"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(sys.argv[1:], args.gpus))   # before anything touches the GPU

    importlib.import_module("automatic-as-built-reconstruction_amd")
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal knobs (1-GPU boxes): AABR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and AABR_BENCH_BACKEND=gloo
    # runs the collective through the host, so the N>1 code path can be exercised where only one GPU exists.
    if os.environ.get("AABR_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("AABR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one device per process: run Function.backward on the calling thread instead of handing every node to the
    # autograd engine's device thread (a torch setting; saves the per-step thread hand-offs, ~1 ms of host time)
    if os.environ.get("AABR_BENCH_AUTOGRAD_THREAD", "0") != "1":
        torch.autograd.set_multithreading_enabled(False)
    if os.environ.get("AABR_BENCH_PIN", "1") != "0":
        pin_host_threads(torch, local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import sparseconvnet as scn
    import dp
    head_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    wl = Workload(scn, torch, dp, dev, head_dtype, rank, world, args.batches)
    # The step's own stream -- forward, backward, update -- runs at HIGH queue priority: the kernels of the critical chain are
    # preferred over the geometry / proposal / weight-gradient streams they share the CUs with.  Measured on one box, two
    # runs each (profiles/r06_switches_ab.txt): fp32 12.72 -> 12.67 ms, bf16 unchanged (host-bound); AABR_BENCH_MAIN_PRIORITY=0
    # puts the step back on the default stream
    main_prio = os.environ.get("AABR_BENCH_MAIN_PRIORITY", "1") == "1"
    if main_prio:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    el, tinfo = timed_steps(torch, dist, wl, args.steps, args.warmup, world, dev,
                            min_timed_s=args.min_timed_s, prewarm=not args.no_prewarm)
    n_timed = args.steps

    if rank == 0:
        n_pts = int(wl.batches[0][0].shape[0])
        V0 = int(wl.last[0][0].metadata.input["V"])
        n_prop = [int(b.shape[0]) for b, _ in wl.last[1]]
        n_anch = [int(l[0].shape[0]) for l in wl.last[2]] if wl.last[2] else None
        line = {
            "metric": "scenes/sec (fwd+bwd)", "value": round(world * n_timed * SCENES_PER_STEP / el, 2),
            "unit": "scenes/s", "n_gpus": world, "steps": n_timed, "warmup": args.warmup,
            "ms_per_step": round(el / n_timed * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %s = default FPN_Net (21.2 M parameters), "
                                   "%s @ 2 cm per GPU and step (%d points -> %d voxels), voxel scatter + "
                                   "all rule books rebuilt every step, fwd + bwd + SGD, RPN head + label generation "
                                   "(criterion-6 IoU + the reference Matcher) + cross-scale top-2000 decode + rotated-3D "
                                   "NMS per scene%s; the top-down levels below the four consumed maps are computed as "
                                   "the reference computes them (fpn_net.py:181-196); rows of every level in %s order" %
                                   (args.config, "'walls' config" if args.config == 2 else "dense whole-building scene",
                                    "4 x S80k scenes" if args.config == 2 else "1 x S1.5M scene", n_pts, V0,
                                    "" if args.dtype == "f32" else ", bf16 feature storage",
                                    "brick-major (brick grids, csrc/brick.hip)" if wl.net.site_order == "brick"
                                    else "the reference's first-seen (hash grids)"),
                       "site_order": wl.net.site_order,
                       "global_batch": world * SCENES_PER_STEP, "points_per_scene": N_POINTS,
                       "voxel_scale": VOXEL_SCALE, "parallelism": "dp%d" % world,
                       "proposals_per_scene": n_prop, "label_generation": bool(wl.label_generation),
                       "gt_boxes_per_scene": N_GT, "anchors_per_scene": n_anch},
            "timed_region_s": round(el, 3),
            "timing": dict(tinfo, min_timed_s=args.min_timed_s,
                           note="`value` = exactly --steps steps after --warmup steps and the disclosed pre-warm "
                                "(`prewarm_*`: untimed full steps until two consecutive 5-step groups agree within "
                                "3 %); `extended` = a second, >= min_timed_s region of the same steps timed right "
                                "after, as a cross-check"),
        }
        if world > 1:
            line["distributed"] = dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                                       allreduce_wait_ms_p50=round(_pct(wl.flat.wait_ms, 0.5), 3) if wl.flat.wait_ms
                                       else None,
                                       allreduce_bytes=int(wl.flat.flat_grad.numel() * wl.flat.flat_grad.element_size()),
                                       grad_buckets=getattr(wl.flat, "bucket_stats", None),
                                       rank_ms_per_step_min=tinfo.get("rank_ms_per_step_min"),
                                       rank_ms_per_step_max=tinfo.get("rank_ms_per_step_max"))
        if el < MIN_TIMED_S:
            line["timed_region_short"] = True
        if not args.no_extras:
            rf, table = roofline_block(torch, wl, head_dtype, el / n_timed * 1e6)
            line["roofline"] = rf
            line["conv_kernels"] = [{k: v for k, v in r.items() if k not in ("flops_per_launch", "grid_threads")}
                                    for r in table[:8]]
            line["conv_step_us_total"] = round(sum(r["step_us"] for r in table), 1)
            fl = sum(r["flops_per_launch"] * r["calls_per_step"] for r in table if r["kind"] == "fwd")
            us = sum(r["step_us"] for r in table if r["kind"] == "fwd")
            line["conv_fwd_flop_weighted_tflops"] = round(fl / us / 1e6, 2) if us else None
            # voxel scatter (A1+A2): N*(32 + 4*C_in) + V*(4*C_in + 16) algorithmic bytes, on the step's own
            # batch and on one 1.5 M-point scene (BASELINE configs[4]'s size)
            locs, feats = wl.batches[0]
            vs = scatter_block(torch, scn, locs, feats, "the step's batch")
            from sparseconvnet import SCN as SCN_
            native = wl.net.site_order == "brick" and SCN_.brick_scatter
            # (brick-native scatter: the level build's mark / scan kernels run for every level of the pyramid too -- the
            # profile's LARGEST grid of each is the input level's launch; the fills are hipMemsetAsync, not in the sum)
            prefixes = ("k_points_", "k_brick_mark", "k_brick_scan", "k_voxel_mean") if native else ("aabr::k_voxel", "k_voxel")
            tr, src = pmc_kernels_traffic(prefixes, None if args.config == 2 else "bench.py --config 4")
            vs["traffic"], vs["traffic_source"] = tr, src
            if tr:
                vs["traffic_over_algorithmic"] = round(tr / vs["bytes"], 2)
            vs["note"] = ("device_seconds = sites_seconds (brick order: points -> brick grid -> rows / chains; first-seen order: "
                          "fill + insert + numbering) + mean_seconds, launches queued "
                          "behind a busy-wait kernel so that no host gap is in the figure; `seconds`: the whole "
                          "InputLayer call incl. its host read of V")
            line["voxel_scatter"] = vs
            if N_POINTS < 1000000:
                try:
                    import synth_scenes as S_
                    l15, f15 = S_.make_batch(1, 1500000, 0, VOXEL_SCALE)
                    line["voxel_scatter_1p5M"] = scatter_block(torch, scn, torch.as_tensor(l15).to(dev),
                                                               torch.as_tensor(f15).to(dev),
                                                               "one 1.5 M-point scene @ 2 cm (BASELINE configs[4])")
                    del l15, f15
                except Exception as e:  # pragma: no cover
                    line["voxel_scatter_1p5M"] = {"error": repr(e)[:200]}
            try:
                line["stage_rooflines"] = stage_rooflines(torch, scn, wl, table, V0)
            except Exception as e:  # pragma: no cover
                line["stage_rooflines"] = {"error": repr(e)[:300]}
            if wl.label_generation:
                import rpn_glue
                maps = wl.last[0]
                with torch.no_grad():
                    t_lab = hip_time(torch, lambda: rpn_glue.rpn_label_matches(
                        maps, wl.base, wl.strides, float(VOXEL_SCALE), wl.targets[0], LABEL_AUG, 6,
                        batch_size=SCENES_PER_STEP), 1, 10)
                pairs = float(sum(n_anch)) * N_GT
                line["label_generation"] = dict(label_iou_us=round(t_lab * 1e6, 1), pairs=int(pairs),
                                                pairs_per_s=round(pairs / t_lab, 1),
                                                note="4 scenes in one library call: anchors of six maps generated in the "
                                                     "kernel, %d x anchors criterion-6 IoUs + best match and threshold "
                                                     "labels per anchor, timed alone (HIP events); inside the timed "
                                                     "step it runs on the side stream beside the RPN head" % N_GT)
            if world == 1:
                extras = {}
                try:
                    # the other storage type, warmed exactly as the headline is (same resident batches, same warm-up
                    # steps, same settle loop): round 5's driver line had this leg 3 % under its stand-alone figure
                    # because it started timing one resident batch after 5 steps
                    other = torch.float32 if args.dtype == "bf16" else torch.bfloat16
                    wl2 = Workload(scn, torch, dp, dev, other, 0, 1, args.batches)
                    n2 = max(20, min(args.steps, 60))
                    el2, ti2 = timed_steps(torch, dist, wl2, n2, max(args.warmup, 10), 1, dev, min_timed_s=0.0,
                                           prewarm=not args.no_prewarm)
                    name = "bf16" if other == torch.bfloat16 else "f32"
                    rf2, table2 = roofline_block(torch, wl2, other, el2 / n2 * 1e6)
                    fl2 = sum(r["flops_per_launch"] * r["calls_per_step"] for r in table2 if r["kind"] == "fwd")
                    us2 = sum(r["step_us"] for r in table2 if r["kind"] == "fwd")
                    extras[name] = {
                        "ms_per_step": round(el2 / n2 * 1e3, 3), "scenes_per_s": round(n2 * SCENES_PER_STEP / el2, 2),
                        "steps": n2, "workload": "same step with %s feature storage" % str(other).split(".")[1],
                        "step_ms": ti2["step_ms"], "roofline": rf2,
                        "conv_step_us_total": round(sum(r["step_us"] for r in table2), 1),
                        "conv_fwd_flop_weighted_tflops": round(fl2 / us2 / 1e6, 2) if us2 else None,
                        "conv_kernels": [{k: v for k, v in r.items() if k not in ("flops_per_launch", "grid_threads")}
                                         for r in table2[:6]]}
                    del wl2
                except Exception as e:  # pragma: no cover
                    extras["other_dtype_error"] = repr(e)[:200]
                try:
                    # the same step in the OTHER site order (headline brick-major -> the reference's first-seen rows over
                    # hash grids, and the other way round): what the internal row order is worth, measured by whoever
                    # runs this file
                    so = "first_seen" if wl.net.site_order == "brick" else "brick"
                    wl4 = Workload(scn, torch, dp, dev, head_dtype, 0, 1, args.batches, site_order=so)
                    n4 = max(20, min(args.steps, 60))
                    el4, ti4 = timed_steps(torch, dist, wl4, n4, max(args.warmup, 10), 1, dev, min_timed_s=0.0,
                                           prewarm=not args.no_prewarm)
                    extras[so] = {
                        "ms_per_step": round(el4 / n4 * 1e3, 3), "scenes_per_s": round(n4 * SCENES_PER_STEP / el4, 2),
                        "steps": n4, "dtype": args.dtype, "step_ms": ti4["step_ms"],
                        "workload": "same step, same storage type, rows of every level in %s order" %
                                    ("the reference's first-seen (hash grids; the library's default)" if so == "first_seen"
                                     else "brick-major (brick grids)")}
                    del wl4
                except Exception as e:  # pragma: no cover
                    extras["site_order_error"] = repr(e)[:200]
                try:
                    # SECOND line, never the headline: the same step with FPN_Net.prune_unused_levels (opt-in) -- the
                    # top-down stages below the last consumed map (m_ups / m_shortcuts / m_mergeds of scales 3..0, which the
                    # reference computes and nobody reads, fpn_net.py:181-196) are not run; returned maps bit-identical
                    # (tests/test_gpu_fpn.py::test_prune_unused_levels_is_bit_identical_and_opt_in)
                    wl.net.prune_unused_levels = True
                    n3 = max(20, min(args.steps, 60))
                    el3, ti3 = timed_steps(torch, dist, wl, n3, 5, 1, dev, min_timed_s=0.0, prewarm=not args.no_prewarm)
                    rf3, table3 = roofline_block(torch, wl, head_dtype, el3 / n3 * 1e6)
                    extras["pruned"] = {
                        "ms_per_step": round(el3 / n3 * 1e3, 3), "scenes_per_s": round(n3 * SCENES_PER_STEP / el3, 2),
                        "steps": n3, "dtype": args.dtype,
                        "workload": "same step, FPN_Net.prune_unused_levels = True (opt-in; NOT the reference's work: its "
                                    "forward_fpn also computes the top-down levels below the four consumed maps)",
                        "conv_step_us_total": round(sum(r["step_us"] for r in table3), 1),
                        "conv_step_us_removed": round(sum(r["step_us"] for r in table) - sum(r["step_us"] for r in table3), 1)}
                except Exception as e:  # pragma: no cover
                    extras["pruned_error"] = repr(e)[:200]
                finally:
                    wl.net.prune_unused_levels = False
                line["extras"] = extras
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, torch)
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

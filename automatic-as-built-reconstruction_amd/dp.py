"""Data-parallel plumbing for the hot path: scenes shard by rank (one process per GPU), the
only exchange step is the gradient all-reduce (RCCL over xGMI through torch.distributed's
"nccl" backend; "gloo" on CPU for tests).

Replaces the reference's DistributedDataParallel wrapper (tools/train_net_sparse3d.py:64-69,
`broadcast_buffers=False`: BatchNorm statistics stay per process) and adds the sharded scene
assignment the reference lacks (data3d/data.py:39-40 has no DistributedSampler).  All
parameters and gradients of the module live in two flat buffers so a step issues ONE
all-reduce (~85 MB fp32 for the full FPN_Net; xGMI ring is per-link bound, so one large
message beats many small buckets) and ONE fused SGD update."""
import time

import torch
import torch.distributed as dist


class FlatParams(object):
    """All parameters of the given modules in ONE flat buffer (views), so a step issues one
    fused SGD update and -- for N > 1 -- one gradient all-reduce.  Gradients are left where
    autograd puts them (no zero-fill and no accumulate launches per step); they are packed into
    the flat gradient buffer by a single multi-tensor copy right before the collective."""

    def __init__(self, modules):
        params = []
        for m in modules:
            params += [p for p in m.parameters() if p.requires_grad]
        self.params = params
        n = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        self.flat = torch.empty(n, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(n, device=dev, dtype=dt)
        self.grad_views = []
        self.wait_ms = []
        o = 0
        for p in params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p.data)
            self.grad_views.append(self.flat_grad[o:o + k].view_as(p.data))
            o += k

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def pack_grads(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.grad_views, grads)

    def allreduce_mean(self, world_size=None, group=None):
        """gradient all-reduce (mean) -- the one collective of a training step"""
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if world_size > 1:
            self.pack_grads()
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
            self.flat_grad.mul_(1.0 / world_size)

    def start_allreduce(self, group=None):
        """Pack the gradients and launch the gradient all-reduce WITHOUT waiting for it: the collective runs
        on the backend's communication stream while the caller enqueues work that does not read the
        parameters (the next scene's hash grid and rule books).  `finish_update` must run before the
        parameters are used again."""
        assert getattr(self, "_pending", None) is None, "previous all-reduce not finished"
        self.pack_grads()
        self._pending = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish_update(self, lr, world_size):
        """wait (stream-ordered for nccl) for the pending all-reduce and apply the SGD update with the mean
        gradient; no-op when nothing is pending"""
        work = getattr(self, "_pending", None)
        if work is None:
            return False
        t0 = time.perf_counter()
        work.wait()
        # host time spent in wait(): ~0 for nccl (stream-ordered wait), the collective's remaining time for gloo
        self.wait_ms.append((time.perf_counter() - t0) * 1e3)
        del self.wait_ms[:-512]
        self._pending = None
        self.flat.add_(self.flat_grad, alpha=-lr / world_size)
        return True

    def sgd_step(self, lr, world_size=1):
        """fused SGD update.  With one rank the gradients are consumed where autograd left them (one
        multi-tensor launch, no packing); with several ranks the all-reduced flat buffer is used."""
        if world_size > 1:
            self.flat.add_(self.flat_grad, alpha=-lr)
        else:
            ps = [p.data for p in self.params if p.grad is not None]
            gs = [p.grad for p in self.params if p.grad is not None]
            if ps:
                torch._foreach_add_(ps, gs, alpha=-lr)

    # ---- bucketed form: collectives start while backward is still running ------------------------------------------
    def begin_bucketed(self, group=None):
        """Arm the compiled backward's hook (sparseconvnet/planExecutor.py `on_grads_ready`): every piece of the
        backward list hands over the slice of the pass's gradient buffer it has completed; the slice is copied into a
        bucket buffer of this object and THAT buffer's all-reduce is launched at once, underneath the remaining
        backward kernels -- what the reference gets from DistributedDataParallel's buckets
        (tools/train_net_sparse3d.py:64-69).  The pass's own gradient buffer is never written by the collective, so
        what autograd stores in `p.grad` stays the LOCAL gradient until `finish_bucketed` replaces it with the mean.
        Call before `backward()`; `finish_bucketed` after it (or `abort_bucketed` when backward raised).

        All ranks must hand over the same pieces: the sizes seen in the first bucketed step are agreed across the
        ranks (`_agree_plan`, one small all-gather, once) and every later piece is checked against them BEFORE its
        collective is launched -- a rank whose graph fell back to the per-module path raises instead of hanging the
        others in a mismatched collective."""
        from sparseconvnet import planExecutor
        self._bk = dict(works=[], params=[], grads=[], bufs=[], group=group, bytes=0, launched_early=0, sizes=[])
        plan = getattr(self, "_bucket_plan", None)

        def ready(piece, n_pieces, flat, pairs):
            bk = self._bk
            if plan is not None and (len(bk["sizes"]) >= len(plan) or plan[len(bk["sizes"])] != flat.numel()):
                raise RuntimeError("bucketed all-reduce: piece %d of %d has %d elements, the plan agreed across the "
                                   "ranks says %s" % (piece, n_pieces, flat.numel(), plan))
            bk["sizes"].append(flat.numel())
            buf = flat.clone()           # main stream; the collective below is ordered behind it
            bk["bufs"].append(buf)
            bk["works"].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True))
            bk["bytes"] += buf.numel() * buf.element_size()
            bk["launched_early"] += 1 if piece < n_pieces - 1 else 0
            base = flat.data_ptr()
            for p_, g_ in pairs:
                o = (g_.data_ptr() - base) // g_.element_size()
                bk["params"].append(p_)
                bk["grads"].append(buf[o:o + g_.numel()].view_as(g_))

        planExecutor.on_grads_ready = ready

    def abort_bucketed(self):
        """disarm the hook after a failed backward; collectives already launched are waited for"""
        from sparseconvnet import planExecutor
        planExecutor.on_grads_ready = None
        bk, self._bk = getattr(self, "_bk", None), None
        if bk:
            for w in bk["works"]:
                try:
                    w.wait()
                except Exception:
                    pass

    def _agree_plan(self, sizes, group):
        """once: every rank must have produced the same list of bucket sizes"""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world > 1:
            mine = torch.zeros(64, dtype=torch.int64, device=self.flat.device)
            mine[0] = len(sizes)
            mine[1:1 + len(sizes)] = torch.tensor(sizes, dtype=torch.int64)
            lo, hi = mine.clone(), mine.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if not (torch.equal(lo, mine) and torch.equal(hi, mine)):
                raise RuntimeError("bucketed all-reduce: the ranks disagree on the bucket plan (mine: %s)" % (sizes,))
        self._bucket_plan = list(sizes)

    def finish_bucketed(self, lr, world_size):
        """after backward(): all-reduce what the hook did not see (parameters outside the compiled graph: the input
        convolution, the RPN head) as one last bucket, wait for every bucket, turn the sums into means, apply the SGD
        update from them and leave `p.grad` = the mean gradient for every parameter.  Returns the number of buckets."""
        from sparseconvnet import planExecutor
        planExecutor.on_grads_ready = None
        bk = self._bk
        plan = getattr(self, "_bucket_plan", None)
        if plan is not None and bk["sizes"] != plan[:-1]:
            self.abort_bucketed()
            raise RuntimeError("bucketed all-reduce: this step produced buckets %s, the agreed plan is %s"
                               % (bk["sizes"], plan))
        seen = {id(p_) for p_ in bk["params"]}
        rest = [p_ for p_ in self.params if id(p_) not in seen and p_.grad is not None]
        n = sum(p_.numel() for p_ in rest)
        if plan is not None and n != plan[-1]:
            self.abort_bucketed()
            raise RuntimeError("bucketed all-reduce: last bucket has %d elements, the agreed plan says %d" % (n, plan[-1]))
        if rest:
            buf = torch.empty(n, device=self.flat.device, dtype=self.flat.dtype)
            views, o = [], 0
            for p_ in rest:
                views.append(buf[o:o + p_.numel()].view_as(p_))
                o += p_.numel()
            torch._foreach_copy_(views, [p_.grad for p_ in rest])
            bk["bufs"].append(buf)
            bk["works"].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=bk["group"], async_op=True))
            bk["bytes"] += n * buf.element_size()
            bk["params"] += rest
            bk["grads"] += views
        if plan is None:
            self._agree_plan(bk["sizes"] + [n], bk["group"])
        t0 = time.perf_counter()
        for w in bk["works"]:
            w.wait()
        self.wait_ms.append((time.perf_counter() - t0) * 1e3)
        del self.wait_ms[:-512]
        if bk["params"]:
            if world_size > 1:
                torch._foreach_mul_(bk["bufs"], 1.0 / world_size)
            torch._foreach_add_([p_.data for p_ in bk["params"]], bk["grads"], alpha=-lr)
            for p_, g_ in zip(bk["params"], bk["grads"]):
                p_.grad = g_             # the mean gradient (clipping, logging and other optimizers read this)
        self.bucket_stats = dict(buckets=len(bk["works"]), launched_during_backward=bk["launched_early"],
                                 bytes=bk["bytes"])
        self._bk = None
        return self.bucket_stats["buckets"]

    def broadcast(self, src=0, group=None):
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat, src, group=group)


def shard_scenes(n_scenes, rank, world_size, sizes=None):
    """Scene indices of this rank.  With `sizes` (points per scene) scenes are dealt in
    descending-size snake order (0..w-1, w-1..0, ...) so every rank gets a similar amount of work
    (DDP waits for the slowest rank)."""
    idx = list(range(n_scenes))
    if sizes is None:
        return idx[rank::world_size]
    idx.sort(key=lambda i: -int(sizes[i]))
    mine = []
    for r0 in range(0, n_scenes, world_size):  # snake order: 0..w-1, w-1..0, ...
        chunk = idx[r0:r0 + world_size]
        j = (world_size - 1 - rank) if (r0 // world_size) % 2 else rank
        if j < len(chunk):
            mine.append(chunk[j])
    return mine

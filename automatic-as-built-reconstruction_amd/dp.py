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

    def __init__(self, modules, grad_dtype=None):
        """`grad_dtype=torch.bfloat16` (option): the gradient MESSAGE of the all-reduce (flat or bucketed) travels in bf16 -- half the
        bytes over xGMI (85 -> 42 MB for FPN_Net; SURVEY 5: "bf16 grads halve the message") -- the local gradients are
        rounded once before the collective, the sum is formed by the backend in bf16 and the mean + update in fp32.
        Default None: the reference's fp32 all-reduce."""
        params = []
        for m in modules:
            params += [p for p in m.parameters() if p.requires_grad]
        self.params = params
        n = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        self.flat = torch.empty(n, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(n, device=dev, dtype=dt)
        self.grad_dtype = grad_dtype if (grad_dtype is not None and grad_dtype != dt) else None
        self.msg = torch.zeros(n, device=dev, dtype=self.grad_dtype) if self.grad_dtype is not None else None
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
            if self.msg is not None:          # reduced-precision message: round once, sum, widen
                self.msg.copy_(self.flat_grad)
                dist.all_reduce(self.msg, op=dist.ReduceOp.SUM, group=group)
                self.flat_grad.copy_(self.msg)
            else:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
            self.flat_grad.mul_(1.0 / world_size)

    def start_allreduce(self, group=None):
        """Pack the gradients and launch the gradient all-reduce WITHOUT waiting for it: the collective runs
        on the backend's communication stream while the caller enqueues work that does not read the
        parameters (the next scene's hash grid and rule books).  `finish_update` must run before the
        parameters are used again."""
        assert getattr(self, "_pending", None) is None, "previous all-reduce not finished"
        self.pack_grads()
        if self.msg is not None:
            self.msg.copy_(self.flat_grad)
        self._pending = dist.all_reduce(self.msg if self.msg is not None else self.flat_grad, op=dist.ReduceOp.SUM,
                                        group=group, async_op=True)

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
        if self.msg is not None:
            self.flat_grad.copy_(self.msg)
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

        All ranks must hand over the same pieces.  In the FIRST bucketed step no plan exists yet, so that step only
        collects the cloned buckets; `finish_bucketed` agrees the list of sizes across the ranks (`_agree_plan`: two tiny
        collectives of fixed shape) and launches the data collectives only then.  From the second step on every piece
        is checked against the agreed plan BEFORE its collective is launched, underneath the backward pass.  Either way
        a rank whose graph fell back to the per-module path (or cut its list differently) raises on every rank instead
        of hanging the others in a mismatched collective -- from step 0 on."""
        from sparseconvnet import planExecutor
        self._bk = dict(works=[], params=[], grads=[], bufs=[], msgs=[], group=group, bytes=0, launched_early=0, sizes=[])
        msg_dtype = self.grad_dtype
        plan = getattr(self, "_bucket_plan", None)

        def ready(piece, n_pieces, flat, pairs):
            bk = self._bk
            if plan is not None and (len(bk["sizes"]) >= len(plan) or plan[len(bk["sizes"])] != flat.numel()):
                raise RuntimeError("bucketed all-reduce: piece %d of %d has %d elements, the plan agreed across the "
                                   "ranks says %s" % (piece, n_pieces, flat.numel(), plan))
            bk["sizes"].append(flat.numel())
            buf = flat.clone()           # main stream; the collective below is ordered behind it
            msg = buf.to(msg_dtype) if msg_dtype is not None else buf    # (option) the message travels in bf16
            bk["bufs"].append(buf)
            bk["msgs"].append(msg)
            if plan is not None:         # agreed plan: launch underneath the rest of the backward pass
                bk["works"].append(dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group, async_op=True))
                bk["launched_early"] += 1 if piece < n_pieces - 1 else 0
            bk["bytes"] += msg.numel() * msg.element_size()
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
        """once, BEFORE the first data collective: every rank must have produced the same list of bucket sizes.  Two
        collectives whose shapes cannot differ between the ranks: the number of buckets (MIN / MAX of one pair), then --
        only when the counts agree -- the sizes themselves (MIN / MAX of a tensor of that agreed length)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world > 1:
            dev = self.flat.device
            cnt = torch.tensor([len(sizes), -len(sizes)], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.MIN, group=group)       # (min count, -max count)
            lo, hi = int(cnt[0]), -int(cnt[1])
            if lo != hi:
                raise RuntimeError("bucketed all-reduce: the ranks disagree on the bucket plan (%d .. %d buckets; mine: "
                                   "%s)" % (lo, hi, sizes))
            mine = torch.tensor(list(sizes) + [-v for v in sizes], dtype=torch.int64, device=dev)
            agreed = mine.clone()
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)    # (min sizes, -max sizes)
            if not torch.equal(agreed, mine):
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
        if plan is None:
            # first bucketed step: nothing has been launched yet (`ready` only collected the buckets) -- agree first
            try:
                self._agree_plan(bk["sizes"] + [n], bk["group"])
            except RuntimeError:
                self.abort_bucketed()
                raise
            for msg in bk["msgs"]:
                bk["works"].append(dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=bk["group"], async_op=True))
        if rest:
            buf = torch.empty(n, device=self.flat.device, dtype=self.flat.dtype)
            views, o = [], 0
            for p_ in rest:
                views.append(buf[o:o + p_.numel()].view_as(p_))
                o += p_.numel()
            torch._foreach_copy_(views, [p_.grad for p_ in rest])
            msg = buf.to(self.grad_dtype) if self.grad_dtype is not None else buf
            bk["bufs"].append(buf)
            bk["msgs"].append(msg)
            bk["works"].append(dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=bk["group"], async_op=True))
            bk["bytes"] += n * msg.element_size()
            bk["params"] += rest
            bk["grads"] += views
        t0 = time.perf_counter()
        for w in bk["works"]:
            w.wait()
        self.wait_ms.append((time.perf_counter() - t0) * 1e3)
        del self.wait_ms[:-512]
        if self.grad_dtype is not None and bk["bufs"]:
            torch._foreach_copy_(bk["bufs"], bk["msgs"])      # widen the summed message back into the fp32 buckets
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


# ---- the two small collectives outside the gradient path ------------------------------------------------------------
def reduce_loss_dict(loss_dict, group=None):
    """The scalar losses of a step averaged onto rank 0 for logging -- `reduce_loss_dict`
    (maskrcnn_benchmark/engine/trainer_sparse3d.py:17-38): keys in sorted order stacked into one tensor, ONE
    `dist.reduce` to rank 0, divided by the world size there only (the other ranks keep the un-normalised partial the
    backend left them, as in the reference).  No-op for one process."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        stacked = torch.stack([torch.as_tensor(loss_dict[k]).detach().reshape(()) for k in names], dim=0)
        # `dst` is a GLOBAL rank: the first member of the group (global rank 0 for the default group)
        dst = dist.get_global_rank(group, 0) if group is not None else 0
        dist.reduce(stacked, dst=dst, group=group)
        if dist.get_rank(group) == 0:
            stacked /= world
        return {k: v for k, v in zip(names, stacked)}


def gather_predictions(predictions_per_rank, group=None):
    """Inference-side gather: every rank holds {scene index: prediction} for the scenes of its shard; rank 0 receives
    the merged list ordered by scene index, the other ranks None -- `_accumulate_predictions_from_multiple_gpus`
    (maskrcnn_benchmark/engine/inference.py:32-51) over `scatter_gather` (utils/comm.py:89-157).  The reference
    pickles each rank's dict to a temporary directory announced through a 256-byte broadcast; here the objects travel
    through `all_gather_object` (RCCL / gloo), no file system in between.  Device tensors inside the predictions are
    moved to the host first (the reference's `torch.save` does the same implicitly)."""
    def to_host(o):
        if torch.is_tensor(o):
            return o.detach().cpu()
        if isinstance(o, dict):
            return {k: to_host(v) for k, v in o.items()}
        if isinstance(o, tuple) and hasattr(o, "_fields"):      # namedtuple: positional fields
            return type(o)(*[to_host(v) for v in o])
        if isinstance(o, (list, tuple)):
            return type(o)(to_host(v) for v in o)
        return o

    mine = to_host(predictions_per_rank)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        parts = [mine]
    else:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, mine, group=group)
        if dist.get_rank(group) != 0:
            return None
    merged = {}
    for p in parts:
        merged.update(p)
    ids = sorted(merged.keys())
    if ids and len(ids) != ids[-1] + 1:
        import logging
        logging.getLogger("aabr.dp").warning("Number of scenes that were gathered from multiple processes is not a "
                                             "contiguous set. Some scenes might be missing from the evaluation")
    return [merged[i] for i in ids]

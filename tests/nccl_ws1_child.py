"""Child process of tests/test_gpu_fpn.py::test_rccl_world_size_1_bucketed_equals_flat.

RCCL readiness without an 8-GPU node: backend `nccl` (= RCCL on ROCm) at world_size 1, the bench's own training step
(bench.Workload: compiled graph, weight gradients on the library's second stream, geometry prefetch and proposals on
side streams), three steps each way from the same state:
  * flat:     backward -> dp.FlatParams.start_allreduce (one RCCL all-reduce of the packed gradients) -> update
  * bucketed: dp.FlatParams.begin_bucketed / finish_bucketed with planExecutor.grad_segments = 4 -- RCCL collectives
              launched from inside the compiled backward pass on slices handed over after aabr_plan_run's side-stream
              join, async work handles, update from the reduced buckets
and the parameters must come out bit-equal.  This exercises RCCL init, its stream and the ordering between the
library's second stream, the hook and the collective, which gloo (host-synchronous) cannot show.
Prints one JSON line.  Started fresh by the test (the parent pytest process has its own GPU context; nothing is
re-exec'd)."""
import importlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    importlib.import_module("automatic-as-built-reconstruction_amd")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.autograd.set_multithreading_enabled(False)
    import bench
    import dp
    import sparseconvnet as scn
    from sparseconvnet import planExecutor
    wl = bench.Workload(scn, torch, dp, dev, torch.float32, 0, 1, 2)
    flat = wl.flat
    start = flat.flat.clone()
    bufs = [b for b in list(wl.net.buffers()) + list(wl.head.buffers())]
    bstart = [b.clone() for b in bufs]
    lr, steps = 1e-2, 3

    def reset():
        torch.cuda.synchronize()
        flat.flat.copy_(start)
        for b, s in zip(bufs, bstart):
            b.copy_(s)

    def run_flat():
        reset()
        for i in range(steps):
            flat.zero_grad()
            wl.forward_backward(i, after_backward=flat.start_allreduce)
            work, flat._pending = flat._pending, None
            work.wait()
            ps = [p.data for p in flat.params]
            torch._foreach_add_(ps, flat.grad_views, alpha=-lr)     # the all-reduced (world 1: unchanged) gradients
        torch.cuda.synchronize()
        return flat.flat.clone()

    def run_bucketed():
        reset()
        n = 0
        for i in range(steps):
            flat.zero_grad()
            planExecutor.grad_segments = 4
            flat.begin_bucketed()
            try:
                wl.forward_backward(i)
            except BaseException:
                flat.abort_bucketed()
                raise
            n = flat.finish_bucketed(lr, 1)
            planExecutor.grad_segments = 0
        torch.cuda.synchronize()
        return flat.flat.clone(), n

    want = run_flat()
    got, n = run_bucketed()
    again, _ = run_bucketed()
    moved = float((want - start).abs().max())
    out = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), equal=bool(torch.equal(want, got)),
               reproducible=bool(torch.equal(got, again)), buckets=n,
               launched_during_backward=flat.bucket_stats["launched_during_backward"],
               bucket_bytes=flat.bucket_stats["bytes"], max_abs_diff=float((want - got).abs().max()),
               max_param_change=moved, hook_disarmed=planExecutor.on_grads_ready is None,
               grads_are_means=bool(sum(p.grad is not None for p in flat.params) > 100))
    print(json.dumps(out))
    sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

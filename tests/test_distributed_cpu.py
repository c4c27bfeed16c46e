"""world_size-2 gloo rehearsal of the N>1 path: scene sharding + the single gradient all-reduce."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import importlib
    importlib.import_module("automatic-as-built-reconstruction_amd")
    import dp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # same init on every rank
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    fp = dp.FlatParams([lin])
    fp.broadcast(0)
    mine = dp.shard_scenes(6, rank, world, sizes=[10, 60, 20, 50, 30, 40])
    fp.zero_grad()
    for s in mine:
        x = torch.full((2, 4), float(s + 1))
        lin(x).sum().backward()
    fp.pack_grads()
    local = fp.flat_grad.clone()
    fp.allreduce_mean(world)
    g_mean = fp.flat_grad.clone()
    fp.sgd_step(0.1, world)
    # the overlapped form used by bench.py: launch the all-reduce, do parameter-independent work, finish
    w_before = fp.flat.clone()
    fp.zero_grad()
    for s in mine:
        lin(torch.full((2, 4), float(s + 2))).sum().backward()
    fp.pack_grads()
    local2 = fp.flat_grad.clone()
    fp.start_allreduce()
    assert fp.finish_update(0.1, world) is True and fp.finish_update(0.1, world) is False
    q.put((rank, mine, local, g_mean, w_before, local2, fp.flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, l0, g0, w0, m0, v0), (_, s1, l1, g1, w1, m1, v1) = res
    assert sorted(s0 + s1) == list(range(6)) and not set(s0) & set(s1)
    # balanced by size: 60+30+20 vs 50+40+10
    assert s0 == [1, 4, 2] and s1 == [3, 5, 0]
    torch.testing.assert_close(g0, (l0 + l1) / 2)
    torch.testing.assert_close(g0, g1)
    torch.testing.assert_close(w0, w1)  # replicas stay in lock-step after the update
    # overlapped all-reduce + update: same result as the mean-gradient SGD step
    torch.testing.assert_close(v0, w0 - 0.1 * (m0 + m1) / 2)
    torch.testing.assert_close(v0, v1)

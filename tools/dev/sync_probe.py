"""Dev probe: how long does a small read-back on a SIDE stream take while the MAIN stream is busy for ~10 ms?
Variants: pageable .tolist(), pinned copy + event.synchronize(), pinned copy + event.query() polling,
pinned copy + stream.synchronize().  Main = torch's default stream (the null stream) or a created one."""
import sys, time
import torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream()
small = torch.arange(16, device=dev, dtype=torch.int32)
pin = torch.empty(16, dtype=torch.int32, pin_memory=True)


def busy(n=12):
    for _ in range(n):
        a @ a


def probe(kind):
    ev0 = torch.cuda.Event()
    ev0.record()
    busy()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        side.wait_event(ev0)
        y = small + 1
        if kind == "tolist":
            r = y.tolist()
        elif kind == "event":
            pin.copy_(y, non_blocking=True); e = torch.cuda.Event(); e.record(); e.synchronize()
        elif kind == "query":
            pin.copy_(y, non_blocking=True); e = torch.cuda.Event(); e.record()
            while not e.query():
                pass
        elif kind == "stream":
            pin.copy_(y, non_blocking=True); side.synchronize()
        elif kind == "kernel_only":
            e = torch.cuda.Event(); e.record(); e.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3


for main_kind in ("default", "created"):
    ctx = torch.cuda.stream(torch.cuda.Stream()) if main_kind == "created" else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        busy(3); torch.cuda.synchronize()
        for kind in ("kernel_only", "tolist", "event", "query", "stream"):
            probe(kind)
            r = [probe(kind) for _ in range(5)]
            print("main=%-8s %-12s read returns after %6.2f ms; main stream busy for %6.2f ms" %
                  (main_kind, kind, sorted(x[0] for x in r)[2], sorted(x[1] for x in r)[2]))

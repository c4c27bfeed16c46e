"""Dev tool: where one bench step (BASELINE configs[2]) spends its HOST time and where the main stream is at each
point: Workload.marks = (name, host clock, HIP event on the main stream) at the phase boundaries of
forward_backward / step.  Per boundary: host time since the step's start, and the main stream's time since the
step's start event when IT reaches the boundary -- host ahead of device = the device has work queued; device at the
boundary right after the host = the host is the one being waited for.   usage: [f32|bf16] [steps]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import bench as B
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), dtype, 0, 1, 2)
for i in range(10):
    wl.step(i)
torch.cuda.synchronize()
wl.marks = []
import rpn_glue
rpn_glue._trace = wl._mark      # the event lands on the proposal stage's stream: when ITS launches are done
t0 = time.perf_counter()
for i in range(n):
    wl.step(i)
torch.cuda.synchronize()
print("wall %.2f ms/step (with the marks' events)" % ((time.perf_counter() - t0) / n * 1e3))
marks = wl.marks
starts = [k for k, m in enumerate(marks) if m[0] == "step start"]
per = len(marks) // n
acc = {}
for s in starts[2:]:
    name0, h0, e0 = marks[s]
    for name, h, e in marks[s:s + per]:
        a = acc.setdefault(name, [0.0, 0.0, 0])
        a[0] += (h - h0) * 1e3
        a[1] += e0.elapsed_time(e)
        a[2] += 1
print("%-36s %10s %10s" % ("boundary", "host ms", "device ms"))
for name, (h, d, c) in acc.items():
    print("%-36s %10.2f %10.2f" % (name, h / c, d / c))

# where does the proposal stage's read-back wait?  events after the stage's last launch, after a trivial kernel, after
# the pinned copy: host times (since the step's start) at which each is first seen complete by polling
if len(sys.argv) > 3 and sys.argv[3] == "rbtrace":
    import _hip
    wl.marks = None
    out = []
    for i in range(14):
        _hip.rb_trace = []
        t0 = time.perf_counter()
        wl.step(i)
        out.append([(k, (t - t0) * 1e3) for k, t in _hip.rb_trace] + [("step returns", (time.perf_counter() - t0) * 1e3)])
    torch.cuda.synchronize()
    for o in out[6:]:
        print(", ".join("%s %.2f" % kv for kv in o))
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "devclock":
    # every boundary = a mailbox post on the CURRENT stream that nobody waits for; its device real-time clock word is
    # read two steps later.  One clock for all streams and no HIP events: the true device-side timeline of the
    # pipelined loop, beside the host's.
    import ctypes as C
    import _hip
    lib = _hip.load()
    one = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    pool, recs, seq = [], [], [0]
    for k in range(64):
        b = C.c_void_p()
        _hip.check(lib.aabr_mailbox_create(64, C.byref(b)))
        pool.append(b.value)

    def mark(name):
        seq[0] += 1
        box = pool[seq[0] % len(pool)]
        _hip.check(lib.aabr_mailbox_post(_hip.ptr(one), 4, box, seq[0], _hip.stream()))
        recs.append([name, time.perf_counter(), box, seq[0], None])
        for r in recs[-40:-24]:          # posts of more than a step ago have run
            if r[4] is None and C.c_uint32.from_address(r[2]).value == r[3]:
                r[4] = C.c_uint32.from_address(r[2] + 4).value

    wl.marks = None
    wl._mark = mark
    rpn_glue._trace = mark
    for i in range(16):
        wl.step(i)
    torch.cuda.synchronize()
    for r in recs:
        if r[4] is None and C.c_uint32.from_address(r[2]).value == r[3]:
            r[4] = C.c_uint32.from_address(r[2] + 4).value
    starts = [k for k, r in enumerate(recs) if r[0] == "step start"]
    h00, d00 = recs[starts[8]][1], recs[starts[8]][4]
    print("%-36s %12s %12s   (ms since the host / the device passed `step start` of the first step shown)" % ("boundary", "host", "device"))
    for r in recs[starts[8]:starts[11]]:
        print("%-36s %12.2f %12s" % (r[0], (r[1] - h00) * 1e3, "%.2f" % (((r[4] - d00) & 0xffffffff) / 1e5) if r[4] is not None else "?"))
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "clock":
    # device real-time clock of the mailbox posts: one on the main stream at the step's start, the proposal stage's own
    import _hip
    wl.marks = None
    clk = []
    orig_rb = _hip.read_back

    def rb(t):
        h0 = time.perf_counter()
        r = orig_rb(t)
        clk.append(("proposal read", _hip.last_post_clock(), h0, time.perf_counter()))
        return r

    rpn_glue._hip.read_back = rb
    one = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    drain = not (len(sys.argv) > 4 and sys.argv[4] == "nodrain")
    import ctypes as C
    boxes = []
    for k in range(2):
        b = C.c_void_p()
        _hip.check(_hip.load().aabr_mailbox_create(64, C.byref(b)))
        boxes.append(b.value)
    pend = None
    for i in range(14):
        h0 = time.perf_counter()
        if drain:
            orig_rb(one)                   # drains the main stream up to here: the step starts on an idle main stream
            clk.append(("step start", _hip.last_post_clock(), h0, time.perf_counter()))
        else:
            # a post on the main stream nobody waits for; its clock word is read one step later (complete by then)
            if pend is not None:
                pend[0][1] = C.c_uint32.from_address(pend[1] + 4).value
            _hip.check(_hip.load().aabr_mailbox_post(_hip.ptr(one), 4, boxes[i & 1], i + 1, _hip.stream()))
            clk.append(["step start", None, h0, h0])
            pend = (clk[-1], boxes[i & 1])
        wl.step(i)
    torch.cuda.synchronize()
    if pend is not None:
        pend[0][1] = C.c_uint32.from_address(pend[1] + 4).value
    torch.cuda.synchronize()
    base = None
    for name, c, h0, h1 in clk[8:]:
        if name == "step start":
            base, hb = c, h1
            print("step start")
        else:
            print("  %s: device ran the post %.2f ms after the step-start post; host issued it at %.2f ms, saw it at %.2f ms"
                  % (name, ((c - base) & 0xffffffff) / 1e5, (h0 - hb) * 1e3, (h1 - hb) * 1e3))
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "readprobe":
    import _hip
    log = []
    step_t0 = [0.0]
    orig = _hip.read_back

    def probe(t):
        evs = []
        for stage in ("last launch", "trivial kernel", "pinned copy"):
            if stage == "trivial kernel":
                t2 = t + 0
            elif stage == "pinned copy":
                host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                host.copy_(t2, non_blocking=True)
            e = torch.cuda.Event()
            e.record()
            evs.append((stage, e))
        seen = {}
        t_in = time.perf_counter()
        while len(seen) < 3:
            for stage, e in evs:
                if stage not in seen and e.query():
                    seen[stage] = time.perf_counter()
        log.append((t_in - step_t0[0], [(s, seen[s] - step_t0[0]) for s, _ in evs]))
        return host.tolist()

    _hip.read_back = probe
    wl.marks = None
    for i in range(12):
        torch.cuda.synchronize()
        step_t0[0] = time.perf_counter()
        wl.step(i)
    torch.cuda.synchronize()
    for t_in, st in log[4:]:
        print("read issued at %.2f ms: " % (t_in * 1e3) + ", ".join("%s done at %.2f" % (s, v * 1e3) for s, v in st))

"""Dev tool: where one bench step (BASELINE configs[2]) spends its time, host side against device side.
At every phase boundary the host clock is read and a HIP event is recorded on the main stream; a phase whose
device completion time tracks the host's enqueue time is host-bound, one that trails it is device-bound.
usage: tools_step_phases.py [f32|bf16] [steps]"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import rpn_glue
import bench as B

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
wl = B.Workload(scn, torch, dp, dev, dtype, 0, 1, 2)
if len(sys.argv) > 3:
    wl.net.compiled_graph = sys.argv[3] != "0"
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()

names = ["net forward", "head + loss", "backward", "prepare(next) on side", "proposals on side", "sgd"]
host = [[] for _ in names]
devt = [[] for _ in names]
tot = []
for i in range(steps):
    torch.cuda.synchronize()
    marks = []

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((time.perf_counter(), e))

    mark()
    wl.flat.zero_grad()
    locs, feats = wl.batches[i % len(wl.batches)]
    rpn_maps, _ = wl.net([locs, feats])
    mark()
    loss, objs, regs = wl.head_loss(rpn_maps)
    ev_fwd = torch.cuda.Event()
    ev_fwd.record()
    mark()
    loss.backward()
    feats.grad = None
    mark()
    main = torch.cuda.current_stream()
    with torch.no_grad():
        wl.net.prepare(wl.batches[(i + 1) % len(wl.batches)], wl.side)
    mark()
    with torch.no_grad(), torch.cuda.stream(wl.side):
        wl.side.wait_event(ev_fwd)
        props = rpn_glue.rpn_proposals(rpn_maps, [o.detach() for o in objs], [r.detach() for r in regs], wl.base,
                                       wl.strides, float(B.VOXEL_SCALE), 2000, 1000, 0.5, (0.3, 0.3))
    main.wait_stream(wl.side)
    mark()
    wl.flat.sgd_step(1e-5, 1)
    mark()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    for k in range(len(names)):
        host[k].append((marks[k + 1][0] - marks[0][0]) * 1e3)
        devt[k].append(marks[0][1].elapsed_time(marks[k + 1][1]))
    tot.append((t_end - marks[0][0]) * 1e3)

print("phase boundaries, ms since step start (median of %d isolated steps; sync before each step, so no overlap "
      "between steps -- the bench loop overlaps the host of step i+1 with the device tail of step i)" % steps)
med = lambda v: sorted(v)[len(v) // 2]
print("%-26s %10s %10s" % ("after", "host", "device"))
for k, n in enumerate(names):
    print("%-26s %10.2f %10.2f" % (n, med(host[k]), med(devt[k])))
print("step wall (isolated): %.2f ms" % med(tot))

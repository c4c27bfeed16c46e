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

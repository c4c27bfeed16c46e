"""Dev probe: wall time per bench step with / without a device-wide synchronize at the end of every step."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import bench as B
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), torch.float32, 0, 1, 2)
for i in range(10):
    wl.step(i)
torch.cuda.synchronize()
for mode in ("pipelined", "sync each step", "pipelined", "sync each step"):
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        wl.step(i)
        if mode != "pipelined":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-16s %.2f ms/step" % (mode, (time.perf_counter() - t0) / n * 1e3))

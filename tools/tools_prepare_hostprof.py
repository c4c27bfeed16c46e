"""Dev tool: cProfile of FPN_Net.prepare (next batch's geometry) and rpn_proposals on the bench workload."""
import cProfile, importlib, io, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import bench as B
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), torch.float32, 0, 1, 2)
for i in range(4):
    wl.step(i)
torch.cuda.synchronize()
side = wl.side
n = 20
t0 = time.perf_counter()
for i in range(n):
    with torch.no_grad():
        wl.net.prepare(wl.batches[i % 2], side)
    wl.net.layers_in[0]._prepared.clear()
torch.cuda.synchronize()
print("prepare: %.2f ms per call (nothing else in flight)" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    with torch.no_grad():
        wl.net.prepare(wl.batches[i % 2], side)
    wl.net.layers_in[0]._prepared.clear()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(45)
print(buf.getvalue()[:9000])
from sparseconvnet import SCN
print("geom plans %d ops %d (over %d prepares)" % (SCN.geom_stats["plans"], SCN.geom_stats["ops"], 2 * n + 4))

"""Dev tool: cProfile of the host side of one bench step (BASELINE configs[2]).  usage: [f32|bf16] [tottime|cumtime]"""
import cProfile, importlib, io, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import bench as B
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
key = sys.argv[2] if len(sys.argv) > 2 else "tottime"
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), dtype, 0, 1, 2)
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    wl.step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print("host loop %.2f ms/step, wall %.2f ms/step" % (t_enq / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    wl.step(i)
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
print(buf.getvalue()[:9000])

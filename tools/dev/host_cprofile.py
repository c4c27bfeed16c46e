"""Dev tool: cProfile of the bench step's host side (single-threaded autograd as in bench.py).  usage: [f32|bf16] [steps]"""
import cProfile, importlib, io, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import dp
import bench as B

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.autograd.set_multithreading_enabled(False)
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), dtype, 0, 1, 2)
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream(device=torch.device("cuda", 0), priority=-1))
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(8, 8 + n):
    wl.step(i)
pr.disable()
torch.cuda.synchronize()
for key, m in (("tottime", 70), ("cumulative", 90)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(m)
    txt = s.getvalue()
    print("=" * 20, key, "(all times are totals over %d steps: divide by %d)" % (n, n))
    print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)

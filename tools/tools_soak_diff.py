"""Dev tool: run n un-synchronised bench steps, then print one checksum per parameter gradient / parameter (to diff
two runs and see WHICH tensors diverge)."""
import hashlib, importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import bench, dp
n = int(sys.argv[1]); dtype = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
wl = bench.Workload(scn, torch, dp, torch.device("cuda", 0), dtype, 0, 1, 3)
for i in range(n):
    wl.step(i)
torch.cuda.synchronize()
names = {id(p): k for k, p in list(wl.net.named_parameters()) + list(wl.head.named_parameters())}
for p in wl.flat.params:
    g = p.grad
    print(names[id(p)], hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest()[:8] if g is not None else "-",
          hashlib.sha256(p.data.cpu().numpy().tobytes()).hexdigest()[:8])
for k, b in wl.net.named_buffers():
    print("buf", k, hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()[:8])

"""Dev tool: time the default 9-scale FPN_Net (fwd, fwd+bwd) on a synthetic scene."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch, numpy as np
import sparseconvnet as scn
import synth_scenes as S
from test_cabi_and_host import default_fpn

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
vs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fdt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = "cuda:0"
torch.manual_seed(0)
net = default_fpn(feature_dtype=fdt).to(dev)
# one device: Function.backward on the calling thread (the engine's device thread costs a hand-off per root;
# with six roots handed in the un-threaded 6.1 ms reads 12 ms)
torch.autograd.set_multithreading_enabled(False)
net.compiled_graph = os.environ.get("AABR_COMPILED_GRAPH", "1") != "0"   # planExecutor (one launch list per pass)
locs, feats = S.make_batch(bs, npts, 0, vs)
l, f = torch.as_tensor(locs).to(dev), torch.as_tensor(feats).to(dev)


grads = None


def run(bwd):
    """bwd: 0 = forward only; 1 = forward + a loss over the six RPN maps + backward (18 + ~30 small torch launches
    of the loss itself); 2 = forward + backward with the output gradients handed in (FPN_Net alone)"""
    global grads
    scn.forward_pass_multiplyAdd_count = 0
    rpn, roi = net([l, f])
    if bwd == 1:
        loss = sum(m.features.square().mean() for m in rpn)
        loss.backward()
    elif bwd == 2:
        if grads is None:
            grads = [torch.randn_like(m.features) * 1e-3 for m in rpn]
        torch.autograd.backward([m.features for m in rpn], grads)
    return rpn


for _ in range(3):
    run(1)
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()      # full collections cost 40-65 ms each in this process: one of them inside a 30-iteration loop reads as +2 ms/iter
f.requires_grad_(True)
for bwd in (0, 1, 2):
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        r = run(bwd)
    torch.cuda.synchronize()
    print("FPN_Net[%s] %d pts x bs%d @scale %d  %s: %.2f ms/iter  (V0=%d, MACs=%.3g)" % (
        str(fdt).split(".")[-1], npts, bs, vs, ("fwd", "fwd+loss+bwd", "fwd+bwd (gradients handed in)")[bwd],
        (time.perf_counter() - t0) / n * 1e3,
        r[0].metadata.input["V"], float(scn.forward_pass_multiplyAdd_count)))
print("max mem GB", torch.cuda.max_memory_allocated() / 1e9)

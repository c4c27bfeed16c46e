"""Dev tool: BatchNormLeakyReLU forward / backward on the [rows, planes] shapes of the bench step (BASELINE
configs[2]): microseconds and achieved HBM rate against the algorithmic bytes (forward: 2 reads + 1 write,
backward: 5 reads + 1 write of the matrix)."""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn

dev = "cuda:0"
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
es = 2 if dtype == torch.bfloat16 else 4


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot_f = tot_b = 0.0
for rows, planes, count in ((309589, 32, 5), (281622, 64, 5), (200652, 64, 5), (200652, 128, 1), (84077, 128, 6),
                            (22250, 128, 6), (5565, 128, 6), (1382, 256, 5), (332, 256, 5), (49, 256, 3)):
    bn = scn.BatchNormLeakyReLU(planes).to(dev)
    x = torch.randn(rows, planes, device=dev).to(dtype).requires_grad_(True)
    t = scn.SparseConvNetTensor(features=x)
    y = bn(t).features
    g = torch.randn_like(y)
    tf = timeit(lambda: bn(t))
    y = bn(t).features
    tb = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    by = rows * planes * es
    print("rows %7d planes %3d: fwd %6.1f us %5.0f GB/s | bwd %6.1f us %5.0f GB/s   (x%d per step)" %
          (rows, planes, tf, 3 * by / tf / 1e3, tb, 6 * by / tb / 1e3, count))
    tot_f += tf * count
    tot_b += tb * count
print("weighted per step: fwd %.0f us, bwd %.0f us" % (tot_f, tot_b))

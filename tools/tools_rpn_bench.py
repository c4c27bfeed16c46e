"""Dev tool: time the device-resident RPN glue (sigmoid -> top-k -> anchors+decode -> rotated NMS) per feature map."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
importlib.import_module("automatic-as-built-reconstruction_amd")
import numpy as np, torch
import sparseconvnet as scn
import rpn_glue
dev = "cuda:0"
rng = np.random.default_rng(0)
for (sx, sy, sz, batch, nsites) in ((64, 64, 8, 1, 6000), (128, 128, 16, 4, 40000)):
    coords = np.unique(np.stack([rng.integers(0, sx, nsites), rng.integers(0, sy, nsites), rng.integers(0, sz, nsites),
                                 rng.integers(0, batch, nsites)], 1), axis=0)
    coords = coords[np.argsort(coords[:, 3], kind="stable")]
    x = scn.InputLayer(3, [sx, sy, sz], mode=4)([torch.as_tensor(coords).to(dev), torch.randn(coords.shape[0], 4, device=dev)])
    V, A = x.features.shape[0], 2
    base = torch.tensor([[0, 0, 0, 0.1, 1.5, 2.5, 0.0], [0, 0, 0, 0.1, 1.5, 2.5, 1.5708]], device=dev)
    obj = torch.randn(V * A, device=dev)
    reg = torch.randn(V * A, 7, device=dev) * 0.1

    def run():
        return rpn_glue.rpn_proposals_single_map(x, obj, reg, base, 20.0, (1.0, 1.0, 1.0), 2000, 1000, 0.5)

    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    print("map %dx%dx%d batch %d: %d sites x %d anchors -> %s proposals: %.1f us per call (%.1f us per example)" % (
        sx, sy, sz, batch, V, A, [int(o[0].shape[0]) for o in out], (time.perf_counter() - t0) / n * 1e6,
        (time.perf_counter() - t0) / n * 1e6 / batch))

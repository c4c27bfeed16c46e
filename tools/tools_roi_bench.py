"""Dev tool: time SparseToDense + ROIAlignRotated3D (forward, backward) at detector-like sizes."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
importlib.import_module("automatic-as-built-reconstruction_amd")
import numpy as np, torch
import sparseconvnet as scn
from maskrcnn_benchmark.layers.roi_align_rotated_3d import ROIAlignRotated3D
dev = "cuda:0"
rng = np.random.default_rng(0)
for (sx, sy, sz, C, nsites, nroi) in ((64, 64, 8, 128, 6000, 512), (128, 128, 16, 128, 20000, 512)):
    coords = np.unique(np.stack([rng.integers(0, sx, nsites), rng.integers(0, sy, nsites), rng.integers(0, sz, nsites),
                                 np.zeros(nsites, np.int64)], 1), axis=0)
    feats = torch.randn(coords.shape[0], C, device=dev, requires_grad=True)
    x = scn.InputLayer(3, [sx, sy, sz], mode=4)([torch.as_tensor(coords).to(dev), feats])
    rois = np.stack([np.zeros(nroi), rng.uniform(4, sx - 4, nroi), rng.uniform(4, sy - 4, nroi), rng.uniform(1, sz - 1, nroi),
                     rng.uniform(2, 12, nroi), rng.uniform(2, 12, nroi), rng.uniform(1, 4, nroi),
                     rng.uniform(-90, 90, nroi)], 1).astype(np.float32)
    r = torch.as_tensor(rois).to(dev)
    layer = ROIAlignRotated3D((7, 7, 3), 1.0, 2)

    def fwd():
        return layer(x, r)

    def fwdbwd():
        feats.grad = None
        x2 = scn.InputLayer(3, [sx, sy, sz], mode=4)([torch.as_tensor(coords).to(dev), feats])
        out = layer(x2, r)
        out.sum().backward()

    for fn, name in ((fwd, "forward"), (fwdbwd, "forward+backward (incl. InputLayer)")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print("grid %dx%dx%d C=%d sites=%d rois=%d  %s: %.1f us" % (sx, sy, sz, C, coords.shape[0], nroi, name,
                                                                  (time.perf_counter() - t0) / n * 1e6))

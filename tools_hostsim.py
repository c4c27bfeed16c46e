"""Dev tool: measure the pure host-side (Python + autograd + ctypes) cost of a bench step with
every kernel-launching C entry point replaced by a no-op, on CPU tensors (no GPU needed)."""
import cProfile, ctypes, importlib, io, os, pstats, sys, time
REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import _hip
import synth_scenes as S

real = _hip.load()
V_FAKE = 64


class Fake(object):
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name.endswith("_floats") or name.endswith("_words") or name in ("aabr_version", "aabr_last_error"):
            return fn
        if name == "aabr_input_layer_sites":
            def sites(*a):
                meta = a[11]
                arr = (ctypes.c_int32 * 8).from_address(meta if isinstance(meta, int) else meta.value)
                arr[0] = V_FAKE; arr[1] = 6; arr[2] = 0
                return 0
            return sites
        return lambda *a: 0


_hip._lib = Fake()
_hip._gpu_ok = True
_hip.require_gpu = lambda t=None: None
_hip.stream = lambda: None
_hip._raw_stream = lambda i: 0
torch._C._cuda_getDevice = lambda: 0
import sparseconvnet as scn
import bench
import dp

# the async read-back needs CUDA events: use the synchronous path
dev = torch.device("cpu")
m = bench.build_model(scn, dev)
flat = dp.FlatParams([v for k, v in m.items() if k != "inp"])
locs, feats = S.make_batch(1, 200, 0, 20)
locs, feats = torch.as_tensor(locs), torch.as_tensor(feats).requires_grad_(True)
g = torch.zeros(V_FAKE, 32)

import sparseconvnet.SCN as SCN
_orig_enq = SCN.Metadata_3.inputLayerEnqueue
class _Ev(object):
    def record(self): pass
    def synchronize(self): pass
torch.cuda.Event = lambda *a, **k: _Ev()
_orig_empty = torch.empty
def _empty(*a, **k):
    k.pop("pin_memory", None)
    return _orig_empty(*a, **k)
torch.empty = _empty


def step():
    flat.zero_grad()
    out = bench.forward(scn, m, locs, feats)
    out.features.backward(g)
    feats.grad = None
    flat.allreduce_mean(1)
    flat.sgd_step(1e-4)


for _ in range(20):
    step()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step()
print("host us/step: %.1f" % ((time.perf_counter() - t0) / N * 1e6))
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N):
        step()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sys.argv[1]).print_stats(40)
    print(s.getvalue()[:9000])

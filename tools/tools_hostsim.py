"""Dev tool: measure the pure host-side (Python + autograd + ctypes) cost of a bench step with
every kernel-launching C entry point replaced by a no-op, on CPU tensors (no GPU needed)."""
import cProfile, ctypes, importlib, io, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
        if name.endswith("_floats") or name.endswith("_words") or name.endswith("_elems") \
                or name in ("aabr_version", "aabr_last_error", "aabr_conv_dw_chunk_pairs"):
            return fn
        if name == "aabr_input_layer_sites":
            def sites(*a):
                meta = a[13]
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
if len(sys.argv) > 1 and sys.argv[1] != "glue":
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N):
        step()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sys.argv[1]).print_stats(40)
    print(s.getvalue()[:9000])

def timeit(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6

with torch.no_grad():
    print("fwd no_grad us: %.1f" % timeit(lambda: bench.forward(scn, m, locs, feats)))
print("fwd grad us: %.1f" % timeit(lambda: bench.forward(scn, m, locs, feats)))
x0 = m["inp"]([locs, feats])
with torch.no_grad():
    print("  inputlayer us: %.1f" % timeit(lambda: m["inp"]([locs, feats])))
    print("  conv1 us: %.1f" % timeit(lambda: m["conv1"](x0)))
    x1 = m["conv1"](x0)
    print("  bn1 us: %.1f" % timeit(lambda: m["bn1"](x1)))
    print("  add us: %.1f" % timeit(lambda: scn.add_feature_planes([x1, x1])))
    import sparseconvnet.SCN as SCN
    w = m["conv2"].weight
    out = torch.empty(0)
    fs = torch.LongTensor([3,3,3])
    print("  SCN.SubmConv_updateOutput us: %.1f" % timeit(lambda: SCN.SubmanifoldConvolution_updateOutput(x1.spatial_size, fs, x1.metadata, x1.features, out, w, torch.Tensor())))
    tb = x1.metadata.getSubmanifoldRuleBook(x1.spatial_size, fs)
    print("  getSubmanifoldRuleBook us: %.1f" % timeit(lambda: x1.metadata.getSubmanifoldRuleBook(x1.spatial_size, fs)))
    print("  _conv_fwd us: %.1f" % timeit(lambda: SCN._conv_fwd(x1.features, out, tb.V_out, tb.out, w, None, 0)))

def fb():
    out = bench.forward(scn, m, locs, feats)
    out.features.backward(g)
print("fwd+bwd us: %.1f" % timeit(fb))
def fb2():
    flat.zero_grad()
    out = bench.forward(scn, m, locs, feats)
    out.features.backward(g)
print("zero+fwd+bwd us: %.1f" % timeit(fb2))
print("pack_grads us: %.1f" % timeit(lambda: flat.pack_grads()))
print("sgd us: %.1f" % timeit(lambda: flat.sgd_step(1e-4)))
x1 = m["conv1"](x0)
d = torch.zeros_like(x1.features)
gw = torch.empty_like(m["conv1"].weight)
fs = torch.LongTensor([3,3,3])
import sparseconvnet.SCN as SCN
print("  SCN.SubmConv_backward us: %.1f" % timeit(lambda: SCN.SubmanifoldConvolution_backward(x0.spatial_size, fs, x0.metadata, x0.features.detach(), torch.empty(0), d, m["conv1"].weight.detach(), gw, torch.Tensor())))

if "glue" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000):
        SCN.SubmanifoldConvolution_backward(x0.spatial_size, fs, x0.metadata, x0.features.detach(), torch.empty(0), d, m["conv1"].weight.detach(), gw, torch.Tensor())
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print(s.getvalue()[:5000])

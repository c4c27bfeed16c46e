"""Dev tool: host wall time per bench step by function (perf_counter wrappers, any thread -- the autograd engine
runs the compiled backward on its own thread, where cProfile does not look).   usage: [f32|bf16] [steps]"""
import importlib, os, sys, time
from collections import defaultdict
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
from sparseconvnet import SCN, planExecutor, fpn_net
import dp, rpn_glue, _hip
import bench as B

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
acc, cnt = defaultdict(float), defaultdict(int)


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            cnt[label] += 1
    setattr(obj, name, g)


lib = _hip.load()


class LibProxy(object):
    """times the C entry points that carry whole passes"""
    def __init__(self, lib):
        self.__dict__["_l"] = lib
        self.__dict__["_c"] = {}

    def __getattr__(self, k):
        c = self._c.get(k)
        if c is None:
            f = getattr(self._l, k)
            if k in ("aabr_plan_run", "aabr_geom_run", "aabr_plan_submit", "aabr_plan_drain"):
                def c(*a, _f=f, _k=k):
                    t = time.perf_counter()
                    r = _f(*a)
                    acc["C " + _k] += time.perf_counter() - t
                    cnt["C " + _k] += 1
                    return r
            else:
                c = f
            self._c[k] = c
        return c


prox = LibProxy(lib)
_hip.load = lambda: prox
wl = B.Workload(scn, torch, dp, torch.device("cuda", 0), dtype, 0, 1, 2)
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
acc.clear(); cnt.clear()
wrap(planExecutor._Pass, "forward", "plan forward (pack + run)")
wrap(planExecutor._Pass, "backward", "plan backward (pack + run)")
wrap(planExecutor._Pass, "__init__", "plan pass init")
wrap(planExecutor, "run_fpn")
wrap(SCN, "compile_streams")
wrap(SCN, "flush_geom")
wrap(SCN.Metadata_3, "buildGridsFromInput")
wrap(SCN.Metadata_3, "inputLayerEnqueue")
wrap(SCN.Metadata_3, "inputLayerFinish")
wrap(fpn_net.FPN_Net, "prepare")
wrap(fpn_net.FPN_Net, "_prebuild_geometry")
wrap(fpn_net.FPN_Net, "_compile_streams", "FPN_Net._compile_streams")
wrap(fpn_net.FPN_Net, "_grids_from_input")
wrap(rpn_glue, "rpn_proposals")
wrap(rpn_glue, "rpn_label_matches")
wrap(wl, "head_loss")
wrap(wl.flat, "sgd_step")
wrap(wl.flat, "zero_grad")
wrap(torch.Tensor, "backward", "loss.backward (all threads)")
wrap(torch.Tensor, "tolist")
wrap(wl.net, "forward", "net.forward")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    wl.step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host loop %.3f ms/step (%d steps, %s)" % ((t1 - t0) / n * 1e3, n, dtype))
import ctypes
st3 = (ctypes.c_int64 * 3)()
lib.aabr_plan_launcher_stats(ctypes.byref(st3, 0), ctypes.byref(st3, 8), ctypes.byref(st3, 16))
print("launcher thread since start: busy %.3f ms, %d parts, %d sleeps (%d steps in total)" % (st3[0] / 1e6, st3[1], st3[2], n + 6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-36s %7.3f ms/step  %6.1f calls/step" % (k, v / n * 1e3, cnt[k] / n))

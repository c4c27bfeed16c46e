"""Dev tool: device time of BatchNorm forward / backward on the coarse scales' small matrices, launched back to back
through aabr_plan_run (no host pacing)."""
import importlib, os, struct, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import _hip
lib = _hip.load()
OP = struct.Struct("<ii6i4f4q12Q")
dev = "cuda:0"
for rows, planes in ((22250, 128), (5565, 128), (1382, 256), (332, 256), (49, 256)):
    x = torch.randn(rows, planes, device=dev)
    y, dy, dx = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
    sm, si = torch.empty(planes, device=dev), torch.empty(planes, device=dev)
    rm, rv = torch.zeros(planes, device=dev), torch.ones(planes, device=dev)
    w, b = torch.ones(planes, device=dev), torch.zeros(planes, device=dev)
    dw, db = torch.empty(planes, device=dev), torch.empty(planes, device=dev)
    ws = torch.empty(int(lib.aabr_bn_scratch_floats(planes)), device=dev)
    f = OP.pack(4, 0, planes, 1, 0, 0, 0, 0, 1e-4, 0.9, 0.0, 0.0, rows, 0, 0, 0, x.data_ptr(), y.data_ptr(),
                sm.data_ptr(), si.data_ptr(), rm.data_ptr(), rv.data_ptr(), w.data_ptr(), b.data_ptr(), ws.data_ptr(),
                0, 0, 0)
    g = OP.pack(5, 0, planes, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, rows, 0, 0, 0, x.data_ptr(), dx.data_ptr(),
                y.data_ptr(), dy.data_ptr(), sm.data_ptr(), si.data_ptr(), w.data_ptr(), dw.data_ptr(), db.data_ptr(),
                ws.data_ptr(), b.data_ptr(), 0)
    for name, rec in (("fwd", f), ("bwd", g)):
        n = 50
        plan = rec * n
        _hip.check(lib.aabr_plan_run(plan, n, _hip.stream()))
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.check(lib.aabr_plan_run(plan, n, _hip.stream()))
        e.record()
        torch.cuda.synchronize()
        print("rows %6d planes %3d %s: %.1f us per call (3 launches)" % (rows, planes, name, a.elapsed_time(e) / n * 1e3))

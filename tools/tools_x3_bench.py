"""Dev tool: the dominant convolution instances of the bench step through the fp32-MFMA kernel, the bf16-storage kernel
and the three-term split kernel (k_conv_cs<.., X3>): microseconds and TFLOP/s (HIP events, groups of launches)."""
import importlib, os, struct, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch, bench, sparseconvnet as scn, dp, _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN
dev = torch.device("cuda", 0); lib = _hip.load()
wl = bench.Workload(scn, torch, dp, dev, torch.float32, 0, 1, 1)
wl.net.compiled_graph = False; SCN.trace = []; wl.forward_backward(0, proposals=False); torch.cuda.synchronize(); tr, SCN.trace = SCN.trace, None
seen = set()
for k, a, b, ga, rows_in, f, d in tr:
    if k != "fwd" or ga.vol != 27:
        continue
    for n_in, n_out in ((128, 128), (64, 64), (64, 128), (256, 256)):
        key = (ga.rows, rows_in, n_in, n_out)
        if key in seen or ga.rows < 20000 or (n_in == 256 and ga.rows > 30000) or (n_in == 64 and ga.rows < 100000):
            continue
        seen.add(key)
        V = ga.rows
        flops = 2.0 * float(ga._ensure_counts().sum().item()) * n_in * n_out if hasattr(ga, "_ensure_counts") else 0
        inp = torch.randn((rows_in, n_in), device=dev); out = torch.empty((V, n_out), device=dev)
        w = torch.randn((27, 1, n_in, n_out), device=dev) * 0.05
        res = {}
        T = lib.aabr_conv_wide_tile_rows(n_in, n_out, rows_in, V, 27)
        if T:
            wp = torch.empty(lib.aabr_conv_wpack_floats(27, n_in, n_out), device=dev)
            check(lib.aabr_conv_pack_weights(ptr(w), 27, n_in, n_out, 0, ptr(wp), stream()))
            bl = ga.blocks_wide(T)
            res["fp32"] = bench.hip_time(torch, lambda: check(lib.aabr_conv_forward_wide(ptr(inp), n_in, rows_in, ptr(out), n_out, V, ptr(bl), T, 27, None, 0, ptr(wp), stream())), 5, 4)
        for form in (0, 2, 3):
            _hip.set_knob("CONV_X3", 1)
            if form:
                _hip.set_knob("X3_FORM", form)
            Tx = lib.aabr_conv_wide_tile_rows_x3(n_in, n_out, rows_in, V, 27)
            if Tx:
                n = int(lib.aabr_conv_wpack_x3_elems(27, n_in, n_out))
                pf = torch.zeros(n, dtype=torch.bfloat16, device=dev); pt = torch.zeros(n, dtype=torch.bfloat16, device=dev)
                rec = struct.pack("<QQQiiiiq", w.data_ptr(), pf.data_ptr(), pt.data_ptr(), 27, n_in, n_out, 2, 0)
                jobs = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
                check(lib.aabr_conv_pack_weights_jobs(ptr(jobs), 1, int(lib.aabr_conv_pack_job_blocks(27, n_in, n_out)), stream()))
                bl = ga.blocks_wide(Tx)
                out2 = torch.empty_like(out)
                t = bench.hip_time(torch, lambda: check(lib.aabr_conv_forward_wide_x3(ptr(inp), n_in, rows_in, ptr(out2), n_out, V, ptr(bl), Tx, 27, None, 0, ptr(pf), None, None, stream())), 5, 4)
                res["x3 form %d %s T=%d" % (form, lib.aabr_conv_last_variant().decode(), Tx)] = t
                if T:
                    err = (out2 - out).abs().max().item() / out.abs().max().item()
                    res["  max |x3 - fp32| / max (form %d)" % form] = err
            _hip.set_knob("X3_FORM", None); _hip.set_knob("CONV_X3", None)
        print("rows %d <- %d, %d -> %d planes, %.2f GFLOP" % (V, rows_in, n_in, n_out, flops / 1e9))
        for name, us in res.items():
            if name.startswith("  "):
                print("   %-44s %.2e" % (name, us))
            else:
                print("   %-44s %8.1f us  %6.1f TFLOP/s" % (name, us * 1e6, flops / us / 1e12))

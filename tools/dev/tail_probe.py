"""Dev probe: k_conv_cs<4,0,1> on prefixes of the dominant rule book (84k rows, 128 -> 128): launch time against the
number of workgroups -- how much of the launch is the last, partly filled round of workgroups (768 slots = 256 CUs x 3)."""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch, bench, sparseconvnet as scn, dp, _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN
dev = torch.device("cuda", 0); lib = _hip.load()
wl = bench.Workload(scn, torch, dp, dev, torch.float32, 0, 1, 1)
wl.net.compiled_graph = False; SCN.trace = []; wl.forward_backward(0, proposals=False); torch.cuda.synchronize(); tr, SCN.trace = SCN.trace, None
ga, rows_in = [(g, r) for k, a, b, g, r, f, d in tr if k == "fwd" and g.vol == 27 and 80000 < g.rows < 90000][0]
n_in = n_out = 128; T = 128; vol = ga.vol; V = ga.rows
inp = torch.randn((rows_in, n_in), device=dev)
w = torch.randn((vol, 1, n_in, n_out), device=dev) * 0.05
wp = torch.empty(lib.aabr_conv_wpack_floats(vol, n_in, n_out), device=dev)
check(lib.aabr_conv_pack_weights(ptr(w), vol, n_in, n_out, 0, ptr(wp), stream()))
table = ga.table.view(vol, V)
for ntile in (192, 384, 480, 576, 657, 768):
    Vp = min(ntile * T, V)
    tp = table[:, :Vp].contiguous()
    words = torch.empty(int(lib.aabr_wide_blocks_words(Vp, vol, T)), dtype=torch.int32, device=dev)
    check(lib.aabr_build_wide_blocks(ptr(tp), Vp, vol, T, ptr(words), stream()))
    out = torch.empty((Vp, n_out), device=dev)
    t = bench.hip_time(torch, lambda: check(lib.aabr_conv_forward_wide(ptr(inp), n_in, rows_in, ptr(out), n_out, Vp, ptr(words), T, vol, None, 0, ptr(wp), stream())), 5, 4)
    nt = (Vp + T - 1) // T
    print("tiles %4d  workgroups %5d = %.2f rounds of 768   %7.1f us   %.3f us per workgroup-slot-round" % (nt, nt * 2, nt * 2 / 768, t * 1e6, t * 1e6 / max(1.0, -(-nt * 2 // 768))))

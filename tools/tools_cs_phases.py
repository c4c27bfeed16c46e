"""Dev tool: per-phase shader clocks of k_conv_cs (DBG variant 4) on the scale-2 rule book of the bench workload.
Needs a `make -C automatic-as-built-reconstruction_amd/csrc DEV=1` build.  usage: tools_cs_phases.py [f32|bf16]"""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch, bench, sparseconvnet as scn, dp, _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN
dev = torch.device("cuda", 0); lib = _hip.load()
wl = bench.Workload(scn, torch, dp, dev, torch.float32, 0, 1, 1)
wl.net.compiled_graph = False; SCN.trace = []; wl.forward_backward(0, proposals=False); torch.cuda.synchronize(); tr, SCN.trace = SCN.trace, None
ga, rows_in = [(g, r) for k, a, b, g, r, f, d in tr if k == "fwd" and g.vol == 27 and 80000 < g.rows < 90000][0]
n_in = n_out = 128
T = lib.aabr_conv_wide_tile_rows(n_in, n_out, rows_in, ga.rows, ga.vol)
inp = torch.randn((rows_in, n_in), device=dev); out = torch.empty((ga.rows, n_out), device=dev)
w = torch.randn((ga.vol, 1, n_in, n_out), device=dev) * 0.05
wpack = torch.empty(lib.aabr_conv_wpack_floats(ga.vol, n_in, n_out), device=dev)
check(lib.aabr_conv_pack_weights(ptr(w), ga.vol, n_in, n_out, 0, ptr(wpack), stream()))
b = ga.blocks_wide(T)
ntile = (ga.rows + T - 1) // T
dbg = torch.zeros((2 * ntile * 4, 5), dtype=torch.int64, device=dev)
for _ in range(3):
    check(lib.aabr_conv_forward_wide(ptr(inp), n_in, rows_in, ptr(out), n_out, ga.rows, ptr(b), T, ga.vol, ptr(dbg), 4 << 8, ptr(wpack), stream()))
torch.cuda.synchronize()
d = dbg.double().cpu()
it = d[:, 4].sum()
print("tile rows", T, "waves", d.shape[0], "iterations/wave avg %.1f" % (it / d.shape[0]))
names = ["issue prefetch", "LDS reads + MFMAs", "accumulate + stage store", "barrier wait"]
tot = d[:, :4].sum()
for i, n in enumerate(names):
    print("%-28s %8.1f clocks/iteration  %5.1f %%" % (n, d[:, i].sum() / it, 100 * d[:, i].sum() / tot))
print("total %.1f clocks/iteration (s_memtime ticks)" % (tot / it))
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    T = lib.aabr_conv_wide_tile_rows_bf16(n_in, n_out, rows_in, ga.rows, ga.vol)
    inb = inp.bfloat16(); outb = torch.empty((ga.rows, n_out), device=dev, dtype=torch.bfloat16)
    n = int(lib.aabr_conv_wpack_bf16_elems(ga.vol, n_in, n_out))
    pf = torch.empty(n, dtype=torch.bfloat16, device=dev); pt = torch.empty_like(pf)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(w), ga.vol, n_in, n_out, ptr(pf), ptr(pt), stream()))
    b = ga.blocks_wide(T)
    ntile = (ga.rows + T - 1) // T
    dbg = torch.zeros((2 * ntile * 4, 5), dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.aabr_conv_forward_wide_bf16(ptr(inb), n_in, rows_in, ptr(outb), n_out, ga.rows, ptr(b), T, ga.vol, ptr(dbg),
                                              4 << 8, ptr(pf), stream()))
    torch.cuda.synchronize()
    d = dbg.double().cpu(); d = d[d[:, 4] > 0]
    it = d[:, 4].sum(); tot = d[:, :4].sum()
    print("bf16: tile rows", T, "waves", d.shape[0], "iterations/wave avg %.1f" % (it / d.shape[0]))
    for i, nme in enumerate(names):
        print("%-28s %8.1f clocks/iteration  %5.1f %%" % (nme, d[:, i].sum() / it, 100 * d[:, i].sum() / tot))
    print("total %.1f clocks/iteration (s_memtime ticks)" % (tot / it))

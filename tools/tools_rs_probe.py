"""Dev tool: the row-stationary bf16 kernel alone on one rule book of the bench workload, with the debug bits of
`flags >> 8` (1: loaders gather nothing, 2: consumers issue no MFMA, 8: loaders idle, 32: no weight loads) -- the bits
exist in a `make -C automatic-as-built-reconstruction_amd/csrc DEV=1` build only.
usage: tools_rs_probe.py [level=3] [n_in=128] [n_out=128]"""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import numpy as np, torch
import _hip, bench, synth_scenes as S
from _hip import ptr, stream, check
from sparseconvnet import SCN
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lib = _hip.load()
_hip.set_knob("CONV_RS", 1)     # the dispatch is off by default (profiles/r03_conv_rs_ab.txt)
dev = torch.device("cuda", 0)
L = []
for j in range(4):
    l, f = S.make_scene(80000, 9000 + j, 50)
    L.append(np.concatenate([l, np.full((l.shape[0], 1), j, np.int64)], 1))
locs = torch.as_tensor(np.concatenate(L, 0)).to(dev)
md = SCN.Metadata_3()
sz = torch.LongTensor([4096, 4096, 512])
md.inputLayer(sz, locs, 4, 4, dev)
two = torch.LongTensor([2, 2, 2])
for _ in range(lvl):
    osz = (sz - two) // two + 1
    md.getRuleBook(sz, osz, two, two)
    sz = osz
tb = md.getSubmanifoldRuleBook(sz, torch.LongTensor([3, 3, 3]))
ga, V, vol = tb.out, tb.V_out, tb.vol
R = sum(ga.rule_counts())
print("level %d: V %d R %d vol %d" % (lvl, V, R, vol))
w = torch.randn((vol, 1, n_in, n_out), device=dev) * 0.05
n = int(lib.aabr_conv_wpack_bf16_elems(vol, n_in, n_out))
pf = torch.empty(n, dtype=torch.bfloat16, device=dev); pt = torch.empty_like(pf)
check(lib.aabr_conv_pack_weights2_bf16(ptr(w), vol, n_in, n_out, ptr(pf), ptr(pt), stream()))
inp = torch.randn((V, n_in), device=dev).bfloat16()
out = torch.empty((V, n_out), device=dev, dtype=torch.bfloat16)
for U in [int(x) for x in os.environ.get("UNITS", "0").split(",")]:
    U = U or lib.aabr_conv_rs_unit_rows(n_in, n_out, V, V, vol)
    words = ga.rs_stream(U)
    nun = (V + U - 1) // U
    hdr = words[:nun * 32].view(nun, 32).cpu().numpy()
    items = 4 * sum(bin(int(x) & 0xf).count("1") for row in hdr for x in row[1:row[0] + 1])
    steps = int(hdr[:, 31].sum())
    offs = int(hdr[:, 0].sum())
    print("U %d: units %d, active (unit, offset) %d, items %d (fill %.3f), steps %d" % (U, nun, offs, items, R / (items * 16.0), steps))
    for dbg in [int(x) for x in os.environ.get("DBGS", "0,1,2,3").split(",")]:
        fn = lambda: check(lib.aabr_conv_forward_rs_bf16(ptr(inp), n_in, V, ptr(out), n_out, V, ptr(words), U, vol, None,
                                                         dbg << 8, ptr(pf), stream()))
        t = bench.hip_time(torch, fn, 4, 6)
        print("   dbg %d: %8.1f us  %7.1f TF" % (dbg, t * 1e6, 2.0 * R * n_in * n_out / t / 1e12))
# phase clocks (dbg 64): per-wave sums of s_memtime deltas, written to the buffer passed as `bias`
U = lib.aabr_conv_rs_unit_rows(n_in, n_out, V, V, vol)
words = ga.rs_stream(U)
nun = (V + U - 1) // U
nwg = (nun + ((nun + 255) // 256) - 1) // ((nun + 255) // 256)
stamps = torch.zeros((nwg, 8, 8), dtype=torch.int64, device=dev)
check(lib.aabr_conv_forward_rs_bf16(ptr(inp), n_in, V, ptr(out), n_out, V, ptr(words), U, vol, ptr(stamps), 64 << 8, ptr(pf), stream()))
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.float64)
c = st[:, 0:4, :].reshape(-1, 8); l = st[:, 4:8, :].reshape(-1, 8)
c = c[c[:, 3] > 0]; l = l[l[:, 4] > 0]
print("phase clocks U=%d (cycles per step, mean over waves): steps/WG mean %.0f max %.0f" % (U, c[:, 3].mean(), c[:, 3].max()))
print("  consumer: reads+MFMA issue %.0f | lgkmcnt wait %.0f | barrier wait %.0f | between steps (offset change, weights, write-out) %.0f" % (
    (c[:, 0] / c[:, 3]).mean(), (c[:, 1] / c[:, 3]).mean(), (c[:, 2] / c[:, 3]).mean(), (c[:, 4] / c[:, 3]).mean()))
print("  loader:   store (waits for its rows) %.0f | gather+descriptor issue %.0f | lgkmcnt wait %.0f | barrier wait %.0f" % (
    (l[:, 0] / l[:, 4]).mean(), (l[:, 1] / l[:, 4]).mean(), (l[:, 2] / l[:, 4]).mean(), (l[:, 3] / l[:, 4]).mean()))

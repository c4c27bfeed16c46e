"""The bench's dominant convolution instance ALONE, so that a rocprofv3 kernel trace of this tool gives it a row of its own
(the per-(kernel, grid) rows of a whole-step trace mix every launch that shares the kernel name and the
grid -- e.g. the 1x1x1 lateral convolution at the same level -- and concurrent streams stretch them).

Builds the bench batch's level-3 submanifold rule book (4 x S80k @ 2 cm, seeds 9000.., brick-major rows: 84,077 rows /
777,725 rules), then launches the 128 -> 128, 3x3x3 forward convolution N times back to back through the library's own
dispatch (aabr_conv_forward_wide / _bf16) and NOTHING else; HIP-event time printed.  Under
    rocprofv3 --kernel-trace -d out -o run -- python3 tools/tools_dominant_instance.py f32 200
`tools/rocpd_stats.py out/.../run.db stats.csv --last 200` is that instance's launch-duration distribution.
usage: [f32|bf16] [N]"""
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch

import synth_scenes as S
import _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN

dev = torch.device("cuda:0")
lib = _hip.load()
bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
order = os.environ.get("AABR_BENCH_SITE_ORDER", "brick")
l, _ = S.make_batch(4, 80000, 9000, 50)
md = SCN.Metadata_3(order)
sizes = [(4096 >> k, 4096 >> k, 512 >> k) for k in range(4)]
md.inputLayer(torch.LongTensor(sizes[0]), torch.as_tensor(l).to(dev), 4, 4, dev)
two = torch.LongTensor([2, 2, 2])
for k in range(3):
    md.getRuleBook(torch.LongTensor(sizes[k]), torch.LongTensor(sizes[k + 1]), two, two)
ga = md.getSubmanifoldRuleBook(torch.LongTensor(sizes[3]), torch.LongTensor([3, 3, 3])).out
V, vol, n_in, n_out = ga.rows, ga.vol, 128, 128
R = int((ga.table >= 0).sum().item())
torch.manual_seed(0)
W = torch.randn((vol, 1, n_in, n_out), device=dev) * 0.05
x = torch.randn((V, n_in), device=dev)
T = SCN.wide_tile_rows(n_in, n_out, V, V, vol, bf)
assert T, "the wide kernel does not take this launch"
blocks = ga.blocks_wide(T)
SCN.flush_geom()
if bf:
    x = x.bfloat16()
    n = int(lib.aabr_conv_wpack_bf16_elems(vol, n_in, n_out))
    pf, pt = torch.empty(n, dtype=torch.bfloat16, device=dev), torch.empty(n, dtype=torch.bfloat16, device=dev)
    check(lib.aabr_conv_pack_weights2_bf16(ptr(W), vol, n_in, n_out, ptr(pf), ptr(pt), stream()))
    out = torch.empty((V, n_out), dtype=torch.bfloat16, device=dev)
    fn = lambda: check(lib.aabr_conv_forward_wide_bf16(ptr(x), n_in, V, ptr(out), n_out, V, ptr(blocks), T, vol, None, 0, ptr(pf), stream()))
else:
    pf = torch.empty(int(lib.aabr_conv_wpack_floats(vol, n_in, n_out)), device=dev)
    check(lib.aabr_conv_pack_weights(ptr(W), vol, n_in, n_out, 0, ptr(pf), stream()))
    out = torch.empty((V, n_out), device=dev)
    fn = lambda: check(lib.aabr_conv_forward_wide(ptr(x), n_in, V, ptr(out), n_out, V, ptr(blocks), T, vol, None, 0, ptr(pf), stream()))
fn()
torch.cuda.synchronize()
variant = lib.aabr_conv_last_variant().decode()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(N):
    fn()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / N * 1e3
fl = 2.0 * R * n_in * n_out
peak = 2516.6 if bf else 157.3
print("%s fwd %d->%d vol %d, %d rows / %d rules (%s rows), tile rows %d: %d launches, %.1f us each (HIP events), %.2f GFLOP -> "
      "%.1f TFLOP/s = %.3f of the %s MFMA peak (%.1f)" % (variant, n_in, n_out, vol, V, R, order, T, N, us, fl / 1e9,
                                                         fl / us / 1e6, fl / us / 1e6 / peak, "bf16" if bf else "fp32", peak))

"""Dev tool: time aabr_conv_forward / backward_weight for several channel counts on the S80k rule book."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch
import sparseconvnet as scn
import synth_scenes as S
import _hip
from _hip import ptr, stream, check

dev = "cuda:0"
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
vs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
locs, feats = S.make_batch(1, npts, 0, vs)
layer = scn.InputLayer(3, list(S.FULL_SCALE), mode=4)
x = layer([torch.as_tensor(locs).to(dev), torch.as_tensor(feats).to(dev)])
tb = x.metadata.getSubmanifoldRuleBook(x.spatial_size, torch.LongTensor([3, 3, 3]))
V = tb.V_out
R = tb.total_rules()
lib = _hip.load()
blocks, pairs = tb.out.blocks(), tb.out.pairs()
print("V=%d R=%d" % (V, R))
for ci, co in ((9, 32), (32, 32), (32, 64), (64, 32), (64, 64), (128, 128), (256, 256), (32, 128), (128, 32)):
    inp = torch.randn(V, ci, device=dev)
    out = torch.empty(V, co, device=dev)
    w = torch.randn(27, ci, co, device=dev)
    wpack = torch.empty(lib.aabr_conv_wpack_floats(27, ci, co), device=dev)
    check(lib.aabr_conv_forward(ptr(inp), ci, V, ptr(out), co, V, ptr(blocks), 27, ptr(w), None, 0, ptr(wpack), stream()))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        check(lib.aabr_conv_forward(ptr(inp), ci, V, ptr(out), co, V, ptr(blocks), 27, ptr(w), None, 4, ptr(wpack), stream()))
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / n * 1e-3
    extra = []
    for dbg in (1, 2, 3, 4):
        a.record()
        for _ in range(n):
            check(lib.aabr_conv_forward(ptr(inp), ci, V, ptr(out), co, V, ptr(blocks), 27, ptr(w), None, 4 | (dbg << 8), ptr(wpack), stream()))
        b.record(); torch.cuda.synchronize()
        extra.append(a.elapsed_time(b) / n * 1e3)
    print("   fwd variants (flat kernel): no-MFMA %.1f us, no-gather %.1f us, neither %.1f us, no main loop %.1f us" % tuple(extra))
    mc = tb.out.max_chunks(ci, co)
    dW = torch.empty_like(w)
    scratch = torch.empty(lib.aabr_conv_dw_scratch_floats(mc, ci, co), device=dev)
    dout = torch.randn(V, co, device=dev)
    check(lib.aabr_conv_backward_weight(ptr(inp), ci, ptr(dout), co, V, ptr(pairs), 27, mc, ptr(dW), None, ptr(scratch), stream()))
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        check(lib.aabr_conv_backward_weight(ptr(inp), ci, ptr(dout), co, V, ptr(pairs), 27, mc, ptr(dW), None, ptr(scratch), stream()))
    b.record(); torch.cuda.synchronize()
    t2 = a.elapsed_time(b) / n * 1e-3
    fl = 2.0 * R * ci * co
    print("%3d->%3d  fwd %8.1f us %6.2f TFLOP/s (%.1f%% fp32 MFMA peak) | dW %8.1f us %6.2f TFLOP/s" % (
        ci, co, t * 1e6, fl / t / 1e12, fl / t / 1e12 / 157.3 * 100, t2 * 1e6, fl / t2 / 1e12))

print("bf16 feature storage (v_mfma_f32_16x16x32_bf16, dense bf16 MFMA peak 2516 TFLOP/s):")
for ci, co in ((32, 32), (64, 64), (128, 128), (256, 256), (32, 128), (128, 32)):
    inp = torch.randn(V, ci, device=dev).to(torch.bfloat16)
    out = torch.empty(V, co, device=dev, dtype=torch.bfloat16)
    w = torch.randn(27, ci, co, device=dev)
    wpack = torch.empty(lib.aabr_conv_wpack_bf16_elems(27, ci, co), device=dev, dtype=torch.bfloat16)
    check(lib.aabr_conv_forward_bf16(ptr(inp), ci, V, ptr(out), co, V, ptr(blocks), 27, ptr(w), None, 0, ptr(wpack), stream()))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        check(lib.aabr_conv_forward_bf16(ptr(inp), ci, V, ptr(out), co, V, ptr(blocks), 27, ptr(w), None, 4, ptr(wpack), stream()))
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / n * 1e-3
    mc = tb.out.max_chunks(ci, co)
    dW = torch.empty_like(w)
    scratch = torch.empty(lib.aabr_conv_dw_scratch_floats(mc, ci, co), device=dev)
    dout = torch.randn(V, co, device=dev).to(torch.bfloat16)
    check(lib.aabr_conv_backward_weight_bf16(ptr(inp), ci, ptr(dout), co, V, ptr(pairs), 27, mc, ptr(dW), None, ptr(scratch), stream()))
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        check(lib.aabr_conv_backward_weight_bf16(ptr(inp), ci, ptr(dout), co, V, ptr(pairs), 27, mc, ptr(dW), None, ptr(scratch), stream()))
    b.record(); torch.cuda.synchronize()
    t2 = a.elapsed_time(b) / n * 1e-3
    fl = 2.0 * R * ci * co
    gb = (R * ci * 2 + V * co * 2) / 1e9  # algorithmic bytes: one gathered row per rule + the output
    print("%3d->%3d  fwd %8.1f us %7.2f TFLOP/s (%.1f%% bf16 MFMA peak, %.0f GB/s gathered) | dW %8.1f us %6.2f TFLOP/s" % (
        ci, co, t * 1e6, fl / t / 1e12, fl / t / 1e12 / 2516 * 100, gb / t, t2 * 1e6, fl / t2 / 1e12))

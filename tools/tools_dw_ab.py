"""A/B of the weight-gradient kernels on the bench's own rule books (4 x S80k @ 2 cm): the 64 x 64-block kernels
(k_conv_dw_pairs[_bf16], one workgroup per chunk and block) against the full-tile kernels (k_conv_dw_full_*, one workgroup
per equal range of the concatenated pair list, whole 128 x 128 blocks) at several workgroup counts; device time of the
whole aabr_conv_backward_weight call (kernel + reduce) with the host taken out.
usage: [f32|bf16] [first_seen|brick]"""
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch

import bench
import synth_scenes as S
import _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN

dev = torch.device("cuda:0")
lib = _hip.load()
bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
order = sys.argv[2] if len(sys.argv) > 2 else "brick"
l, _ = S.make_batch(4, 80000, 9000, 50)
md = SCN.Metadata_3(order)
sizes = [(4096 >> k, 4096 >> k, 512 >> k) for k in range(9)]
md.inputLayer(torch.LongTensor(sizes[0]), torch.as_tensor(l).to(dev), 4, 4, dev)
three, two = torch.LongTensor([3, 3, 3]), torch.LongTensor([2, 2, 2])
for k in range(7):
    md.getRuleBook(torch.LongTensor(sizes[k]), torch.LongTensor(sizes[k + 1]), two, two)
dt = torch.bfloat16 if bf else torch.float32
fn = lib.aabr_conv_backward_weight_bf16 if bf else lib.aabr_conv_backward_weight
WGS = [0, 128, 256, 384, 512, 768, 1024, 1536]


def run(name, ga, n_in, n_out, rows_in=None):
    vol, V = ga.vol, ga.rows
    torch.manual_seed(1)
    rows_in = V if rows_in is None else rows_in          # the gather's partner rows index the INPUT matrix
    assert int(ga.table.max()) < rows_in
    x = torch.randn((rows_in, n_in), device=dev).to(dt)
    g = torch.randn((V, n_out), device=dev).to(dt)
    pairs = ga.pairs()
    mc = ga.max_chunks(n_in, n_out)
    SCN.flush_geom()
    R = int(sum(ga.rule_counts()))
    mc = ga.max_chunks(n_in, n_out)     # (the exact chunk count now that the counts are on the host)
    scratch = torch.empty(int(lib.aabr_conv_dw_scratch_floats(mc, n_in, n_out)), device=dev)
    dW = torch.empty((vol, n_in, n_out), device=dev)
    call = lambda: check(fn(ptr(x), n_in, ptr(g), n_out, V, ptr(pairs), vol, mc, ptr(dW), None, ptr(scratch), stream()))
    _hip.set_knob("DW_FULL", 0)
    call()
    ref = dW.clone()
    t0 = bench.device_time(torch, call)
    _hip.set_knob("DW_FULL", None)
    out = ["%-18s %7d rows %8d rules %3d->%-3d chunks %5d | blocks %6.1f us %6.1f TF |" % (
        name, V, R, n_in, n_out, mc, t0 * 1e6, 2.0 * R * n_in * n_out / t0 / 1e12)]
    for w in WGS:
        _hip.set_knob("DW_FULL_WGS", w if w else None)
        _hip.set_knob("DW_FULL_MIN", 1 if not w else None)
        dW.fill_(float("nan"))
        call()
        v = lib.aabr_conv_last_variant().decode()
        if "full" not in v:
            out.append(" %s:-" % (w or "dflt"))
            continue
        err = float((dW - ref).abs().max() / ref.abs().max())
        t = bench.device_time(torch, call)
        out.append(" %s:%.1f" % (w or "dflt", t * 1e6) + ("" if err < 1e-4 else "(err %.1e)" % err))
    _hip.set_knob("DW_FULL_WGS", None)
    _hip.set_knob("DW_FULL_MIN", None)
    print("".join(out), flush=True)


print("weight gradient, %s storage, site order %s; full-tile kernel by number of workgroups (us per call)" % ("bf16" if bf else "fp32", order))
for k in (3, 2, 4, 1, 5, 0):
    tb = md.getSubmanifoldRuleBook(torch.LongTensor(sizes[k]), three)
    run("subm L%d 128->128" % k, tb.out, 128, 128)
tb = md.getSubmanifoldRuleBook(torch.LongTensor(sizes[6]), three)
run("subm L6 256->256", tb.out, 256, 256)
tb = md.getSubmanifoldRuleBook(torch.LongTensor(sizes[5]), three)
run("subm L5 256->256", tb.out, 256, 256)
one = torch.LongTensor([1, 1, 1])
for k in (3, 1):
    tb = md.getSubmanifoldRuleBook(torch.LongTensor(sizes[k]), one)
    run("1x1x1 L%d 128->128" % k, tb.out, 128, 128)
for k in (1, 3):
    tb = md.getRuleBook(torch.LongTensor(sizes[k]), torch.LongTensor(sizes[k + 1]), two, two)
    run("down L%d->%d 128->128" % (k, k + 1), tb.out, 128, 128, rows_in=tb.V_in)
    run("up   L%d->%d 128->128" % (k + 1, k), tb.inn, 128, 128, rows_in=tb.V_out)

"""A/B of the geometry builders: hash grids (first-seen order) against brick grids (brick-major order), device time with
the host taken out (bench.device_time), on the bench's 4-scene batch (configs[2], ~310 k sites) and on one 1.5 M-point
scene (configs[4], ~890 k sites).  Prints one table; `profiles/r05_brick_geometry_ab.txt` is its output."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import numpy as np
import torch

import bench
import synth_scenes as S
import _hip
from _hip import ptr, stream, check
from sparseconvnet import SCN

dev = torch.device("cuda:0")
SP = torch.LongTensor([4096, 4096, 512])
lib = _hip.load()


def scene(kind):
    if kind == "4xS80k":
        l, _ = S.make_batch(4, 80000, 9000, 50)
    else:
        l, _ = S.make_batch(1, 1500000, 0, 50, ext=(40.0, 30.0, 2.7))
    return torch.as_tensor(l).to(dev)


def time_input(locs, order):
    keep = []

    def fn():
        md = SCN.Metadata_3(order)
        md.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
        keep.append(md)
        del keep[:-4]
    return bench.device_time(torch, fn)


def run(kind):
    locs = scene(kind)
    rows = []
    res = {}
    for order in ("first_seen", "brick"):
        md = SCN.Metadata_3(order)
        V = md.inputLayer(SP, locs, 4, 4, dev)
        g = md.grids[(4096, 4096, 512)]
        t_sites = time_input(locs, "first_seen")             # the voxel scatter itself (same in both)
        t_brickify, t_native = 0.0, 0.0
        if order == "brick":
            # the scatter straight into the brick grid (no hash table): prepare + level build from the points + rows / chains
            md0 = SCN.Metadata_3("brick")
            md0.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
            ext = md0._pending["meta"].tolist()[8:12]
            keep0 = []

            def nfn():
                mdn = SCN.Metadata_3("brick")
                mdn.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
                keep0.append((mdn, mdn._brick_scatter_launch(mdn._pending["piece"], mdn.input["n"], mdn.input["spatial"], ext, dev)))
                del keep0[:-4]
            t_native = bench.device_time(torch, nfn)
        if order == "brick":
            # brickify alone: rebuild the brick level + renumber from the first-seen sites
            mdf = SCN.Metadata_3("first_seen")
            mdf.inputLayerEnqueue(SP, locs, 4, dev, asynchronous=False)
            pend = mdf._pending
            m = pend["meta"].tolist()
            keep = []

            def bfn():
                md2 = SCN.Metadata_3("brick")
                md2.input = dict(mdf.input)
                md2._brick_rows = None
                keep.append(md2._brickify_input(pend, m[0], m[8:12]))
                keep.append(md2)
                del keep[:-8]
            t_brickify = bench.device_time(torch, bfn)
        table_ = torch.empty((27, g.V), dtype=torch.int32, device=dev)
        counts = torch.empty(27 * ((g.V + 255) // 256), dtype=torch.int32, device=dev)
        fs = _hip.i32x3((3, 3, 3))
        if order == "brick":
            bk = g.brick

            def subm():
                check(lib.aabr_brick_submanifold_table(ptr(g.coords), g.V, bk.dims_c(), bk.dir_ptr(), bk.bricks_ptr(), fs,
                                                       ptr(table_), ptr(counts), stream()))
        else:
            def subm():
                check(lib.aabr_submanifold_table(ptr(g.coords), g.V, ptr(g.keys), g.cap, fs, ptr(table_), ptr(counts),
                                                 stream()))
        t_subm = bench.device_time(torch, subm)
        R = int((table_ >= 0).sum().item())
        # the strided pyramid of FPN_Net: 8 levels of 2/2 + the four z-collapse grids
        sizes = [(4096 >> k, 4096 >> k, 512 >> k) for k in range(9)]

        def pyramid():
            md3 = SCN.Metadata_3(order)
            md3.input = md.input
            md3.input_spatial = md.input_spatial
            md3.grids = {sizes[0]: g}
            md3._brick_rows, md3._brick_nrows = md._brick_rows, 1
            if order == "brick":
                specs = [(sizes[k + 1], sizes[k], (2, 2, 2), (2, 2, 2)) for k in range(8)]
                specs += [((s[0], s[1], 1), s, (1, 1, s[2]), (1, 1, s[2])) for s in sizes[4:8]]
                md3.buildBrickPyramid(specs)
            else:
                for base in (0, 4):
                    specs = [(torch.LongTensor(sizes[k]), torch.LongTensor([1 << (k - base)] * 3))
                             for k in range(base + 1, min(base + 4, 8) + 1)]
                    for k in range(max(base, 4), min(base + 4, 8)):
                        s = sizes[k]
                        specs.append((torch.LongTensor([s[0], s[1], 1]),
                                      torch.LongTensor([1 << (k - base), 1 << (k - base), (1 << (k - base)) * s[2]])))
                    md3.buildGridsFromInput(torch.LongTensor(sizes[base]), specs)
            return md3
        md3 = pyramid()
        Vs = [md3.grids[s].V for s in sizes]
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pyramid()
        torch.cuda.synchronize()
        t_pyr = (time.perf_counter() - t0) / 5
        # strided tables level 0 -> 1
        g1 = md3.grids[sizes[1]]
        t_out = torch.empty((8, g1.V), dtype=torch.int32, device=dev)
        t_in = torch.empty((8, g.V), dtype=torch.int32, device=dev)
        c1 = torch.empty(8 * ((g1.V + 255) // 256), dtype=torch.int32, device=dev)
        c2 = torch.empty(8 * ((g.V + 255) // 256), dtype=torch.int32, device=dev)
        two = _hip.i32x3((2, 2, 2))
        osz = _hip.i32x3(sizes[1])
        if order == "brick":
            def tabs():
                check(lib.aabr_brick_convolution_tables(ptr(g.coords), g.V, g.brick.dims_c(), g.brick.dir_ptr(),
                                                        g.brick.bricks_ptr(), ptr(g1.coords), g1.V, g1.brick.dims_c(),
                                                        g1.brick.dir_ptr(), g1.brick.bricks_ptr(), two, two, osz,
                                                        ptr(t_out), ptr(t_in), ptr(c1), ptr(c2), stream()))
        else:
            def tabs():
                check(lib.aabr_convolution_tables2(ptr(g.coords), g.V, ptr(g.keys), g.cap, ptr(g1.coords), g1.V,
                                                   ptr(g1.keys), g1.cap, two, two, osz, ptr(t_out), ptr(t_in), ptr(c1),
                                                   ptr(c2), stream()))
        t_tabs = bench.device_time(torch, tabs)
        res[order] = dict(V=V, R=R, Vs=Vs, sites_us=t_sites * 1e6, brickify_us=t_brickify * 1e6, native_us=t_native * 1e6,
                          subm_us=t_subm * 1e6,
                          pyramid_wall_us=t_pyr * 1e6, tables01_us=t_tabs * 1e6,
                          bricks=int(md3.grids[sizes[0]].brick.meta[1].item()) if order == "brick" else 0)
    a, b = res["first_seen"], res["brick"]
    assert a["V"] == b["V"] and a["R"] == b["R"] and a["Vs"] == b["Vs"], (a, b)
    by = 16 * a["V"] + 4 * 27 * a["V"]
    print("%s: %d points, %d sites, %d rules (k=3), %d bricks (%.1f sites per brick); levels %s" %
          (kind, locs.shape[0], a["V"], a["R"], b["bricks"], a["V"] / max(b["bricks"], 1), a["Vs"]))
    print("  %-46s %12s %12s" % ("device time (us), host taken out", "hash grid", "brick grid"))
    print("  %-46s %12.1f %12.1f" % ("voxel scatter, geometry half", a["sites_us"], b["sites_us"]))
    print("  %-46s %12s %12.1f" % ("  + brick level + renumbering of the input sites", "-", b["brickify_us"]))
    print("  %-46s %12s %12.1f   (replaces the two rows above)" % ("voxel scatter straight into the brick grid", "-", b["native_us"]))
    print("  %-46s %12.1f %12.1f   (%.0f / %.0f GB/s of 16 V + 108 V bytes)" %
          ("submanifold rule table 3x3x3, input level", a["subm_us"], b["subm_us"], by / a["subm_us"] / 1e3,
           by / b["subm_us"] / 1e3))
    print("  %-46s %12.1f %12.1f" % ("strided tables level 0 <-> 1 (2/2)", a["tables01_us"], b["tables01_us"]))
    print("  %-46s %12.1f %12.1f   (3 reads / 1 read)" % ("12 strided grids: wall time incl. host reads", a["pyramid_wall_us"],
                                                         b["pyramid_wall_us"]))


if __name__ == "__main__":
    for kind in sys.argv[1:] or ["4xS80k", "S1.5M"]:
        run(kind)

"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header
declares, the host logic (module surface, spatial-size arithmetic, parameter inventory) matches
the reference's contract, operators fail loudly without a GPU, and the kernels' IoU arithmetic
(compiled for the host from the same header) agrees with the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import oracle_lib as O
import synth_scenes as S

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import _hip
    hdr = open(os.path.join(REPO, "include", "aabr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(aabr_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = C.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libaabr_hip.so does not export %s" % name
    assert set(_hip.EXPORTED_SYMBOLS) == declared  # the Python binding covers the whole header
    assert _hip.load().aabr_version() >= 100


def test_argument_validation_without_gpu():
    """entry points validate before touching the device (no compute happens here)"""
    import _hip
    lib = _hip.load()
    assert lib.aabr_input_layer_sites(None, -1, 3, None, 64, None, None, None, None, None, None, None, None,
                                      None, None) == -1
    assert b"ncols" in lib.aabr_last_error()
    assert lib.aabr_conv_forward(None, 0, 10, None, 4, 10, None, 27, None, None, 0, None, None) == -1
    assert lib.aabr_conv_wpack_floats(27, 32, 32) == 27 * 1 * 2 * 512
    assert lib.aabr_conv_wpack_floats(27, 9, 32) == 27 * 1 * 2 * 512
    assert lib.aabr_bn_scratch_floats(32) > 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_operators_fail_loudly_without_gpu():
    import sparseconvnet as scn
    import _hip
    layer = scn.InputLayer(3, [64, 64, 64], mode=4)
    with pytest.raises(_hip.AabrError):
        layer([torch.zeros(4, 3, dtype=torch.long), torch.zeros(4, 2)])
    from second.core.non_max_suppression.nms_gpu import rotate_iou_gpu_eval
    with pytest.raises((_hip.AabrError, RuntimeError, AssertionError)):
        rotate_iou_gpu_eval(np.zeros((2, 5), np.float32), np.zeros((2, 5), np.float32))


def test_operator_surface_matches_reference_names():
    import sparseconvnet as scn
    for name in ("InputLayer", "SubmanifoldConvolution", "Convolution", "Deconvolution", "BatchNormalization",
                 "BatchNormReLU", "BatchNormLeakyReLU", "Sequential", "ConcatTable", "AddTable", "JoinTable",
                 "Identity", "NetworkInNetwork", "SparseConvNetTensor", "Metadata", "add_feature_planes", "FPN_Net"):
        assert hasattr(scn, name), name
    for name in ("InputLayer_updateOutput", "InputLayer_updateGradInput", "SubmanifoldConvolution_updateOutput",
                 "SubmanifoldConvolution_backward", "Convolution_updateOutput", "Convolution_backward",
                 "Deconvolution_updateOutput", "Deconvolution_backward", "BatchNormalization_updateOutput",
                 "BatchNormalization_backward", "Metadata_3", "n_rulebook_bits"):
        assert hasattr(scn.SCN, name), name
    assert scn.SCN.n_rulebook_bits() == 32
    from maskrcnn_benchmark.layers import nms  # noqa: F401
    from maskrcnn_benchmark.structures.boxlist_ops_3d import boxlist_nms_3d, boxlist_iou_3d  # noqa: F401
    from second.pytorch.core.box_torch_ops import rotate_nms_3d  # noqa: F401
    from second.core.non_max_suppression.nms_cpu import rotate_nms_3d_cc  # noqa: F401
    from second.core.non_max_suppression.nms_gpu import rotate_iou_gpu_eval  # noqa: F401
    from utils3d.rotate_nms_3d_torch import boxes_iou_3d, iou_one_dim  # noqa: F401


def default_fpn(**extra):
    import sparseconvnet as scn
    return scn.FPN_Net([4096, 4096, 512], 3, ["xyz", "color", "normal"], 1,
                       [32, 64, 64, 128, 128, 128, 256, 256, 256], 128, True, [4, 3, 2, 1], [4, 3, 2, 1],
                       [[[2, 2, 2]] * 8, [[2, 2, 2]] * 8],
                       [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]], [1, 2, 3, 4, 5, 6], leakiness=0,
                       voxel_scale=20, bn_momentum=0.95, **extra)


def test_fpn_net_inventory_matches_reference_probe():
    """SURVEY.md §2.2 / §8 A12 [probe of the reference]: 21,212,660 parameters, 36 SubmConv +
    12 Conv + 8 Deconv + 35 BN, weight layout [vol, groups, nIn, nOut]."""
    import sparseconvnet as scn
    net = default_fpn()
    assert sum(p.numel() for p in net.parameters()) == 21212660
    mods = list(net.modules())
    # layers_in is counted once; layers_out's BN is the 35th
    assert sum(isinstance(m, scn.SubmanifoldConvolution) for m in mods) == 36
    assert sum(isinstance(m, scn.Convolution) for m in mods) == 12
    assert sum(isinstance(m, scn.Deconvolution) for m in mods) == 8
    assert sum(isinstance(m, scn.BatchNormalization) for m in mods) == 35
    assert tuple(net.layers_in[1].weight.shape) == (27, 1, 9, 32)
    assert tuple(net.convs_pro2d[0].weight.shape) == (32, 1, 128, 128)
    names = dict(net.named_parameters())
    assert "m_downs.1.0.1.weight" in names and "m_mergeds.7.weight" in names
    assert "m_downs.0.0.1.0.running_mean" in dict(net.named_buffers())


def test_spatial_size_arithmetic():
    import sparseconvnet as scn
    conv = scn.Convolution(3, 4, 4, [2, 2, 2], [2, 2, 2], False)
    assert conv.input_spatial_size(torch.LongTensor([128, 128, 16])).tolist() == [256, 256, 32]
    dec = scn.Deconvolution(3, 4, 4, [2, 2, 2], [2, 2, 2], False)
    assert dec.input_spatial_size(torch.LongTensor([256, 256, 32])).tolist() == [128, 128, 16]
    # the full-scale chain of the default backbone: 4096,4096,512 halved 8 times
    s = torch.LongTensor([4096, 4096, 512])
    for _ in range(8):
        s = (s - 2) // 2 + 1
    assert s.tolist() == [16, 16, 2]


def test_kernel_iou_arithmetic_on_host_matches_oracle(tmp_path):
    so = str(tmp_path / "libhostiou.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(REPO, "tests", "iou_math_host_harness.cpp")])
    L = C.CDLL(so)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    L.host_iou_eval.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int, f32p]
    b7, _ = S.make_nms_boxes(400, 7)
    b5 = np.ascontiguousarray(b7[:, [0, 1, 3, 4, 6]])
    b5[5] = b5[4]  # an exact duplicate pair
    for crit in (-1, 0, 1, 2, 3, 4, 5, 6):
        got = np.zeros((400, 400), np.float32)
        L.host_iou_eval(b5, 400, b5, 400, crit, got)
        np.testing.assert_array_equal(got, O.rotate_iou_eval(b5, b5, crit))
    g = np.load(os.path.join(REPO, "tests", "golden", "iou_golden.npz"))
    got = np.zeros((4, 4), np.float32)
    L.host_iou_eval(g["t_boxes"], 4, g["t_boxes"], 4, -1, got)
    np.testing.assert_array_equal(got, g["t_iou"])


def test_lazy_mac_counter_ring_logic():
    """host logic of the lazily evaluated multiply-add counter: slots, generations, folding"""
    import sparseconvnet.SCN as SCN
    ring = SCN._TotalsRing(torch.device("cpu"))
    ring.CAP = 8
    ring.buf = torch.zeros(8, dtype=torch.float64)
    acc = SCN.LazyMacs()
    want = 0.0
    slots = []
    for i in range(29):  # wraps the 8-slot ring three times
        gen, idx = ring.alloc()
        ring.buf[idx] = float(i + 1)
        slots.append((ring, gen, idx))
        acc += SCN.LazyMacs([(ring, gen, idx, 2.0)])
        want += 2.0 * (i + 1)
        if i % 5 == 4:  # read in the middle: folds what is pending, keeps counting afterwards
            assert float(acc) == want
    assert float(acc) == want and int(acc) == int(want)
    assert acc == want and acc > want - 1 and acc <= want
    assert "%d" % acc == "%d" % want
    # terms of a retired generation are still readable (the buffer was read back when it wrapped)
    old = SCN.LazyMacs([slots[17] + (1.0,)])
    assert float(old) == 18.0
    # plain numbers mix in
    acc2 = SCN.LazyMacs() + 5
    acc2 += 7.0
    assert float(acc2 + SCN.LazyMacs([slots[28] + (3.0,)])) == 12.0 + 3.0 * 29


def test_dw_chunk_policy_is_consistent_with_the_library():
    import _hip
    lib = _hip.load()
    assert lib.aabr_conv_dw_chunk_pairs(66094, 27, 32, 32) == 256      # S80k rule book, narrow layer
    assert lib.aabr_conv_dw_chunk_pairs(66094, 27, 128, 128) == 1024   # wide layer: fewer, larger partials
    assert lib.aabr_conv_dw_chunk_pairs(1000000, 27, 32, 32) == 1024   # large rule book
    import sparseconvnet.SCN as SCN
    g = SCN._Gather(None, None, 27, 66094)
    assert g.max_chunks(32, 32) == (27 * 66094 + 255) // 256 + 27
    g._host_counts = [300] * 27
    assert g.max_chunks(32, 32) == 27 * 2 and g.max_chunks(256, 256) == 27


def test_cat_scales_obj_reg_regroups_example_major():
    """SURVEY §8(f) rank 1 glue (rpn_sparse3d.py:19-77): [scale][example] row blocks -> [example][scale]"""
    import rpn_glue
    rng = np.random.default_rng(4)
    scopes = [[(0, 5), (5, 12)], [(0, 2), (2, 3)], [(0, 0), (0, 4)]]  # 3 scales x 2 examples (one empty block)
    obj = [torch.as_tensor(rng.standard_normal((1, 1, sc[-1][1], 1)).astype(np.float32)) for sc in scopes]
    reg = [torch.as_tensor(rng.standard_normal((1, 1, sc[-1][1], 7)).astype(np.float32)) for sc in scopes]
    o, r = rpn_glue.cat_scales_obj_reg(obj, reg, scopes)
    assert o.shape == (19, 1) and r.shape == (19, 7)
    want_o, want_r = [], []
    for b in range(2):
        for s in range(3):
            a, e = scopes[s][b]
            want_o.append(obj[s].reshape(-1, 1)[a:e])
            want_r.append(reg[s].reshape(-1, 7)[a:e])
    assert torch.equal(o, torch.cat(want_o)) and torch.equal(r, torch.cat(want_r))


def test_metadata_incremental_input_is_host_bookkeeping_like_the_reference():
    """Metadata.batchAddSample / setInputSpatialLocation(s) (pybind.cpp:15-19; Metadata.cpp:30-44,81-145): a new
    location appends a feature row (first-seen numbering), a repeated one overwrites or is ignored; nothing touches
    the GPU until a geometry query."""
    import sparseconvnet as scn
    md = scn.Metadata(3)
    md.setInputSpatialSize(torch.LongTensor([8, 8, 8]))
    feats = torch.FloatTensor()
    md.batchAddSample()
    md.setInputSpatialLocation(feats, torch.LongTensor([1, 2, 3]), torch.FloatTensor([1, 10]), False)
    md.setInputSpatialLocation(feats, torch.LongTensor([4, 4, 4]), torch.FloatTensor([2, 20]), False)
    md.setInputSpatialLocation(feats, torch.LongTensor([1, 2, 3]), torch.FloatTensor([3, 30]), False)   # ignored
    assert feats.tolist() == [[1, 10], [2, 20]]
    md.setInputSpatialLocation(feats, torch.LongTensor([1, 2, 3]), torch.FloatTensor([4, 40]), True)    # overwritten
    assert feats.tolist() == [[4, 40], [2, 20]]
    md.batchAddSample()
    md.setInputSpatialLocations(feats, torch.LongTensor([[1, 2, 3], [0, 0, 0], [1, 2, 3]]),
                                torch.FloatTensor([[5, 50], [6, 60], [7, 70]]), True)
    assert feats.tolist() == [[4, 40], [2, 20], [7, 70], [6, 60]]       # same location, other sample: new row
    assert md.getNActive(torch.LongTensor([8, 8, 8])) == 4
    md.setInputSpatialLocations(feats, torch.LongTensor([[7, 7, 7, 2]]), torch.FloatTensor([[8, 80]]), False)
    assert feats.shape == (5, 2) and md._inb["nsamples"] == 3


def test_bench_roofline_traffic_lookup_matches_the_committed_pmc_profile():
    """bench.pmc_traffic: the (kernel, grid) instance named by aabr_conv_last_variant() is found in the committed PMC
    profile although the profiler prints every template argument; a different grid or kernel gives (None, reason),
    never another instance's number."""
    import importlib
    import json
    bench = importlib.import_module("bench")
    bench.PMC_PROFILE = "r03_pmc_fetch_write_per_kernel.json"     # the lookup logic, on a profile whose keys are known
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                        "r03_pmc_fetch_write_per_kernel.json")
    pm = json.load(open(path))["kernels"]
    key = "k_conv_cs<4,0,1,false,1,2,false>|grid=336384"
    assert key in pm
    want = int((2.0 * pm[key]["FETCH_SIZE_KB_avg"] + pm[key]["WRITE_SIZE_KB_avg"]) * 1024)
    got, src = bench.pmc_traffic("k_conv_cs<4,0,1>", 336384)
    assert got == want and key in src
    assert bench.pmc_traffic("k_conv_cs<4,0,1>", 12345)[0] is None
    assert bench.pmc_traffic("k_conv_cs<9,9,9>", 336384)[0] is None
    # the bf16-storage instance of the same rule book (64-row tiles x 128-column slabs: the same grid size) is a
    # different entry (template argument `true`, slab argument 2), merged in from the `--dtype bf16` passes
    kb = "k_conv_cs<2,0,1,true,2,2,false>|grid=336384"
    got_b, src_b = bench.pmc_traffic("k_conv_cs<2,0,1,bf16,x128>", 336384)
    assert kb in pm and kb in src_b and got_b != got
    assert bench.pmc_traffic("k_conv_cs<2,0,1,bf16>", 336384)[0] is None           # 64-column slabs: not this instance


def test_bench_pmc_lookups_resolve_in_this_rounds_profile():
    """the profile bench.py names for THIS round is committed and both lookups resolve in it: the dominant convolution
    instance (fp32 and bf16 storage) by (kernel, grid) and the voxel-scatter stage's kernels by name"""
    import importlib
    import json
    bench = importlib.import_module("bench")
    bench.PMC_PROFILE = "r04_pmc_fetch_write_per_kernel.json"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", bench.PMC_PROFILE)
    pm = json.load(open(path))["kernels"]
    grids = sorted(int(k.split("|grid=")[1]) for k in pm if k.startswith("k_conv_cs<4,0,1,false,1,2,false,false"))
    assert grids
    got, src = bench.pmc_traffic("k_conv_cs<4,0,1>", 336384)
    assert got and got > 50e6 and "336384" in src
    got_b, src_b = bench.pmc_traffic("k_conv_cs<2,0,1,bf16,x128>", 336384)
    assert got_b and got_b != got and ",true,2," in src_b
    tr, names = bench.pmc_kernels_traffic(("aabr::k_voxel", "k_voxel"))
    assert tr and 50e6 < tr < 400e6 and "k_voxel_bin_build" in names and "backward" not in names


def test_knob_registry_rejects_unknown_names_and_takes_known_ones():
    """aabr_set_knob: the tuning knobs are a closed list read once per process (never getenv on a launch path)"""
    import _hip
    lib = _hip.load()
    assert lib.aabr_set_knob(b"NO_SUCH_KNOB", 1, 0) != 0
    for gone in (b"CONV_X3", b"CONV_RS", b"WIDE_OCC4", b"WIDE_DEFER", b"WIDE_NW8"):     # removed with their kernels (round 5)
        assert lib.aabr_set_knob(gone, 1, 0) != 0
    assert b"unknown knob" in lib.aabr_last_error()
    for name in ("CONV_WIDE", "WIDE_NBUF", "BN_SMALL", "WIDE_PRIO", "PLAN_SIDE_BATCH", "PLAN_SIDE_PRIO", "VOXEL_MEAN",
                 "WIDE_NCB", "CONV_NARROW", "DW_FULL", "DW_FULL_MIN", "DW_FULL_WGS", "SPLIT_ROWS", "GEOM_JOBS"):
        assert lib.aabr_set_knob(name.encode(), 1, 0) == 0
        assert lib.aabr_set_knob(name.encode(), 0, 1) == 0                       # back to "unset"

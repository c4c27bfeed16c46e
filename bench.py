#!/usr/bin/env python3
"""bench.py -- scenes/s (fwd+bwd) of the sparse-3D hot path on MI355X.

Workload (BASELINE.json configs[1]): one synthetic SUNCG-shaped scene per step (~80k points,
5 cm voxels, C_in = 9, bs = 1) -> HIP voxel scatter (InputLayer mode 4, hash grid + rule table
built on the device every step) -> 2-stage submanifold backbone (SubmConv3 9->32, residual
block BNReLU-SubmConv3-BNReLU-SubmConv3 32->32, add), fp32, forward + backward (all weight
gradients and the input-feature gradient) + gradient all-reduce (N > 1) + SGD update.
Inputs are resident in HBM before the timed region.  One process per GPU; ranks take
different scenes (weak scaling), the only collective is the gradient all-reduce.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the output-stationary
MFMA gather-GEMM, k_conv_blocks_mfma, 32->32 forward instance) timed with HIP events on the
stream it is launched on; `cpu_baseline` is the oracle (CPU port of the reference path) timed
on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth_scenes as S  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec


def build_model(scn, dev, c_in=9):
    torch.manual_seed(0)
    m = dict(inp=scn.InputLayer(3, list(S.FULL_SCALE), mode=4),
             conv1=scn.SubmanifoldConvolution(3, c_in, 32, 3, False).to(dev),
             bn1=scn.BatchNormLeakyReLU(32, momentum=0.95, leakiness=0).to(dev),
             conv2=scn.SubmanifoldConvolution(3, 32, 32, 3, False).to(dev),
             bn2=scn.BatchNormLeakyReLU(32, momentum=0.95, leakiness=0).to(dev),
             conv3=scn.SubmanifoldConvolution(3, 32, 32, 3, False).to(dev))
    return m


def forward(scn, m, locs, feats, after_geometry=None):
    x0 = m["inp"]([locs, feats])
    if after_geometry is not None:
        after_geometry()  # hook between the parameter-free geometry and the first layer with weights
    x1 = m["conv1"](x0)
    x3 = m["conv3"](m["bn2"](m["conv2"](m["bn1"](x1))))
    return scn.add_feature_planes([x1, x3])


def cpu_baseline(n_scenes_budget_s=15.0):
    """oracle (CPU port of the reference path) on the same workload, bounded sample"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    import ref_net
    # the CPU leg runs on the cores the process was given, not on the four the GPU leg pinned itself to
    if _ORIG_AFFINITY:
        try:
            os.sched_setaffinity(0, _ORIG_AFFINITY)
        except Exception:
            pass
    # host cores this process may use (a 1-GPU box exposes a 16-core share of the host)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    O.set_threads(max(1, min(ncpu, int(os.environ.get("AABR_CPU_THREADS", "16")))))
    rng = np.random.default_rng(0)
    W1 = (rng.standard_normal((27, 9, 32)) * 0.09).astype(np.float32)
    W2 = (rng.standard_normal((27, 32, 32)) * 0.05).astype(np.float32)
    W3 = (rng.standard_normal((27, 32, 32)) * 0.05).astype(np.float32)
    bn = dict(weight=np.ones(32, np.float32), bias=np.zeros(32, np.float32), running_mean=np.zeros(32),
              running_var=np.ones(32))
    locs, feats = S.make_batch(1, 80000, 0, 20)
    done, t0 = 0, time.time()
    while True:
        c = ref_net.two_stage_forward(locs, feats, W1, W2, W3, bn, bn)
        ref_net.two_stage_backward(c, np.ones_like(c["out"]), W1, W2, W3, bn, bn)
        done += 1
        el = time.time() - t0
        if el > n_scenes_budget_s or done >= 64:
            break
    return dict(value=round(done / el, 3), unit="scenes/s", cores=O.num_threads(), kind="port",
                sample="%d x S80k@5cm scene, voxel scatter + rule book + 2-stage fwd+bwd, %.1f s of CPU work, "
                       "OpenMP over %d threads" % (done, el, O.num_threads()))


def time_dominant_kernel(scn, m, scene, reps=30):
    """average launch duration of k_conv_blocks_mfma (32->32 forward, S80k rule book) with HIP
    events recorded on the stream the kernel is launched on (torch's current stream)."""
    import _hip
    from _hip import ptr, stream, check
    lib = _hip.load()
    with torch.no_grad():
        x0 = m["inp"]([scene[0], scene[1]])
        y1 = m["bn1"](m["conv1"](x0))
        tb = x0.metadata.getSubmanifoldRuleBook(x0.spatial_size, torch.LongTensor([3, 3, 3]))
        inp = y1.features.contiguous()
        V = inp.size(0)
        out = torch.empty((V, 32), device=inp.device)
        w = m["conv2"].weight.detach().contiguous()
        wpack = torch.empty(lib.aabr_conv_wpack_floats(27, 32, 32), device=inp.device)
        check(lib.aabr_conv_forward(ptr(inp), 32, V, ptr(out), 32, V, ptr(tb.out.blocks()), 27, ptr(w), None, 0, ptr(wpack),
                                    stream()))
        torch.cuda.synchronize()
        # events bracket GROUPS of back-to-back launches of the one kernel: an event pair around a single
        # 20 us launch reads 2-3 us high (the record / completion-signal cost), which a rocprofv3 trace of the
        # same launch does not contain
        group = 8
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            for _ in range(group):
                check(lib.aabr_conv_forward(ptr(inp), 32, V, ptr(out), 32, V, ptr(tb.out.blocks()), 27, ptr(w), None,
                                            4, ptr(wpack), stream()))
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) / group for a, b in evs)
        ms = ms[: max(1, len(ms) * 3 // 4)]  # drop the slow tail (first-touch / clock ramp)
        R = tb.total_rules()
        return sum(ms) / len(ms) * 1e-3, R, V


def time_stage(fn, reps=10):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def fpn_net_extra(scn, dev):
    """Not the headline: the whole FPN_Net backbone (BASELINE.json configs[2]-shaped input: 4 scenes @ 2 cm)
    forward + backward in fp32 and with bf16 feature storage, so the bench output also carries the
    full-network numbers.  Any failure here is reported in the object and never touches the main line."""
    res = {}
    try:
        locs, feats = S.make_batch(4, 80000, 9000, 50)
        l, f = torch.as_tensor(locs).to(dev), torch.as_tensor(feats).to(dev)
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            torch.manual_seed(0)
            net = scn.FPN_Net([4096, 4096, 512], 3, ["xyz", "color", "normal"], 1,
                              [32, 64, 64, 128, 128, 128, 256, 256, 256], 128, True, [4, 3, 2, 1], [4, 3, 2, 1],
                              [[[2, 2, 2]] * 8, [[2, 2, 2]] * 8],
                              [[256, 256, 32], [128, 128, 16], [64, 64, 8], [32, 32, 4]], [1, 2, 3, 4, 5, 6],
                              leakiness=0, voxel_scale=50, bn_momentum=0.95, feature_dtype=dt).to(dev)

            def run():
                scn.forward_pass_multiplyAdd_count = 0
                rpn, _ = net([l, f])
                sum(m_.features.square().mean() for m_ in rpn).backward()
                return rpn
            for _ in range(6):
                r = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                r = run()
            torch.cuda.synchronize()
            dt_s = (time.perf_counter() - t0) / n
            res[name] = {"ms_fwd_bwd": round(dt_s * 1e3, 2), "scenes_per_s": round(4 / dt_s, 1)}
            res["sites"] = int(r[0].metadata.input["V"])
            res["forward_macs"] = float(scn.forward_pass_multiplyAdd_count)
            del net
        res["workload"] = "FPN_Net (21.2 M parameters), 4 x S80k scenes @ 2 cm, forward + backward"
    except Exception as e:  # pragma: no cover
        res["error"] = repr(e)[:200]
    return res


_ORIG_AFFINITY = None


def pin_host_threads(local_rank, ncores=4):
    """Bind this process (launch thread + autograd thread) to a few cores of the GPU's NUMA node.  On a
    2-socket host the unbound process wanders over 256 hardware threads and the launch-bound step time
    moves by 10-30 % from run to run (measured: 1,890-2,010 scenes/s unbound, 2,140-2,190 bound).
    Best effort: any failure leaves the affinity untouched."""
    global _ORIG_AFFINITY
    try:
        allowed = sorted(os.sched_getaffinity(0))
        _ORIG_AFFINITY = set(allowed)
        if len(allowed) <= ncores:
            return None
        cand = allowed
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            bdf = None
        if bdf:
            node_file = "/sys/bus/pci/devices/%s/numa_node" % str(bdf).lower()
            if os.path.exists(node_file):
                node = int(open(node_file).read().strip())
                if node >= 0:
                    cl = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
                    cpus = []
                    for part in cl.split(","):
                        a, _, b = part.partition("-")
                        cpus += list(range(int(a), int(b or a) + 1))
                    cand = [c for c in cpus if c in set(allowed)] or allowed
        # distinct blocks for the ranks sharing a node: stride by rank modulo the blocks that fit
        nblocks = max(1, len(cand) // ncores)
        b0 = (local_rank % nblocks) * ncores
        cores = set(cand[b0:b0 + ncores])
        os.sched_setaffinity(0, cores)
        return sorted(cores)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults sized for steady state: the first few hundred steps of a fresh process run with the GPU
    # and host clocks still ramping (measured on MI355X: 200 timed steps after 20 warm-up steps read
    # ~0.8 ms/step, 2000 after 300 read ~0.55-0.6 ms/step); the whole default run is still ~3 s
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--scenes", type=int, default=4, help="distinct resident scenes per rank, cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-profile", default="", help="write a cProfile of the timed loop to this file")
    ap.add_argument("--prefetch", type=int, default=0,
                    help="1: build the next scene's hash grid on a side stream while the current scene trains")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal knobs (1-GPU boxes): AABR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # AABR_BENCH_BACKEND=gloo runs the collective through the host, so the N>1 code path can be
    # exercised where only one GPU exists.  Never set by the driver.
    if os.environ.get("AABR_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("AABR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if os.environ.get("AABR_BENCH_PIN", "1") != "0":
        pin_host_threads(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import sparseconvnet as scn
    import dp
    m = build_model(scn, dev)
    flat = dp.FlatParams([v for k, v in m.items() if k != "inp"])
    flat.broadcast(0)

    # resident inputs: `scenes` distinct scenes per rank (different seeds on every rank)
    scenes = []
    for i in range(args.scenes):
        locs, feats = S.make_batch(1, 80000, 1000 * rank + i, 20)
        scenes.append((torch.as_tensor(locs).to(dev), torch.as_tensor(feats).to(dev)))
    grads = []
    with torch.no_grad():
        for sc in scenes:
            o = forward(scn, m, sc[0], sc[1])
            g = torch.Generator(device=dev).manual_seed(7)
            grads.append(torch.randn(o.features.shape, device=dev, generator=g))
    feats_req = [(sc[0], sc[1].clone().requires_grad_(True)) for sc in scenes]

    side = torch.cuda.Stream()
    overlap = world > 1 and os.environ.get("AABR_BENCH_OVERLAP", "0") == "1"

    def step(i):
        j = i % len(scenes)
        flat.zero_grad()
        # geometry of the NEXT scene (hash grid + site numbering; needs only its coordinates) is
        # built on a side stream now, so its one host read-back overlaps this scene's compute
        nj = (i + 1) % len(scenes)
        if args.prefetch and i == 0:
            m["inp"].prepare(feats_req[j][0], dev, side)
        # AABR_BENCH_OVERLAP=1 (N > 1, not the default: it could not be measured on a multi-GPU node this
        # round): this scene's geometry does not read the parameters, so it can overlap the previous step's
        # gradient all-reduce; the collective is then waited for (stream-ordered) and the update applied
        # right before the first convolution needs the weights
        hook = (lambda: flat.finish_update(1e-4, world)) if overlap else None
        out = forward(scn, m, feats_req[j][0], feats_req[j][1], hook)
        out.features.backward(grads[j])
        if args.prefetch:
            m["inp"].prepare(feats_req[nj][0], dev, side)
        feats_req[j][1].grad = None
        if overlap:
            flat.start_allreduce()
        else:
            flat.allreduce_mean(world)
            flat.sgd_step(1e-4, world)

    # The full-network extra (N = 1) runs BEFORE the headline loop: it is required output either way, and half a
    # second of real work ahead of the warm-up steps means the timed region does not start on idle clocks when
    # the caller asks for a short --warmup.
    fpn_extra = None
    if not args.no_cpu_baseline and world == 1:
        fpn_extra = fpn_net_extra(scn, dev)
    else:
        # same purpose where the extra is not produced (N > 1, --no-cpu-baseline): ~0.4 s of untimed forward
        # passes on every rank, so clocks are up before the W warm-up steps whatever W is
        t_pre = time.perf_counter()
        with torch.no_grad():
            while time.perf_counter() - t_pre < 0.4:
                for sc in scenes:
                    forward(scn, m, sc[0], sc[1])
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    flat.finish_update(1e-4, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof = None
    if args.host_profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    flat.finish_update(1e-4, world)  # the last step's update belongs to the timed region
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(45)
        open(args.host_profile, "w").write("enqueue s/step: %g\n" % (t_enq / args.steps) + buf.getvalue())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        ksec, R, V = time_dominant_kernel(scn, m, scenes[0])
        flops = 2.0 * R * 32 * 32
        # HBM traffic of that kernel from the committed PMC passes (profiles/, separate --pmc runs of
        # this same command): FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE
        traffic = None
        try:
            pm = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_fetch_write_per_kernel.json")))["kernels"]
            for kname, v in pm.items():
                if "k_conv_blocks_mfma_buf<2, 4" in kname:  # <NBW, WPB[, ADJ]>
                    traffic = int((2.0 * v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024)
        except Exception:
            traffic = None
        roof = dict(kernel="k_conv_blocks_mfma_buf<2,4,true,true> (SubmConv3 32->32 forward)", bound="mfma",
                    achieved=round(flops / ksec / 1e12, 4), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(flops / ksec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 5), traffic=traffic,
                    launch_us=round(ksec * 1e6, 2), rules=int(R), sites=int(V),
                    algorithmic_flops_per_launch=flops)
        # voxel scatter (A1+A2): N*(32 + 4*C_in) + V*(4*C_in + 16) algorithmic bytes
        N = scenes[0][0].shape[0]
        with torch.no_grad():
            t_sc = time_stage(lambda: m["inp"]([scenes[0][0], scenes[0][1]]))
        sc_bytes = N * (32 + 4 * 9) + V * (4 * 9 + 16)
        scatter = dict(bytes=sc_bytes, seconds=round(t_sc, 7), achieved_gbs=round(sc_bytes / t_sc / 1e9, 2),
                       frac_of_hbm_peak=round(sc_bytes / t_sc / 1e9 / PEAK_HBM_GBS, 5),
                       note="whole InputLayer call incl. hash build, site numbering, host read of V")
        line = {
            "metric": "scenes/sec (fwd+bwd)", "value": round(world * args.steps / el, 2), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S80k@5cm scene (79,998 pts -> ~66k voxels), voxel scatter + 2-stage "
                                   "SubmanifoldConvolution backbone (9->32, residual 32->32 x2), fwd+bwd+SGD, "
                                   "bs=1 per GPU (BASELINE.json configs[1])",
                       "global_batch": world, "points_per_scene": int(N), "parallelism": "dp%d" % world},
            "roofline": roof, "voxel_scatter": scatter,
        }
        if not args.no_cpu_baseline and world == 1:
            line["fpn_net"] = fpn_extra
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

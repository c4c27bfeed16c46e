"""A/B: how much do the convolution kernels gain when the site order is spatially coherent?  Runs bench.py's
dominant-kernel table on the configs[2] workload with the generator's (random within a surface) point order and
with the points of every scene sorted by 8-voxel cell."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag, env in (("random", {}), ("sorted", {"AABR_BENCH_SORT_POINTS": "1"})):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       env=dict(os.environ, **env), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stderr[-1500:]); continue
    d = json.loads(line[0])
    print(tag, "ms/step", d["ms_per_step"], "conv_step_us_total", d["conv_step_us_total"], "bf16", d.get("extras"))
    for k in d["conv_kernels"]:
        print("   ", k["kind"], k["kernel"], "%d->%d" % (k["n_in"], k["n_out"]), "rows", k["rows_out"], "R", k["rules"],
              "x%d" % k["calls_per_step"], k["launch_us"], "us", k["tflops"], "TF")

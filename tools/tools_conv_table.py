"""Dev tool: every convolution launch instance of one bench step (fwd / input-gradient / dW), timed alone."""
import importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
importlib.import_module("automatic-as-built-reconstruction_amd")
import torch, bench, sparseconvnet as scn, dp
dev = torch.device("cuda", 0)
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
wl = bench.Workload(scn, torch, dp, dev, dt, 0, 1, 1)
for i in range(2):
    wl.step(i)
mr = int(os.environ["AABR_TABLE_MAX_ROWS"]) if os.environ.get("AABR_TABLE_MAX_ROWS") else None   # coarse scales only
rows = bench.conv_kernel_table(torch, wl, dt, mr)
tot = sum(r["step_us"] for r in rows)
print("total conv us/step %.0f over %d instances" % (tot, len(rows)))
for r in rows:
    print("%-4s %-44s %4d->%-4d vol %2d rows %7d R %8d x%d %8.1f us  %6.1f TF  step %7.1f" % (
        r["kind"], r["kernel"], r["n_in"], r["n_out"], r["vol"], r["rows_out"], r["rules"], r["calls_per_step"],
        r["launch_us"], r["tflops"], r["step_us"]))

#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_round4.sh [pmc]
# The round's measurement set: (1) rocprofv3 kernel trace of the driver's bench command -> per-kernel and per-(kernel,
# grid) stats; (2) the un-profiled default line and the driver-command line; (3) the --config 4 line (1.5 M points, bf16);
# with `pmc`: (4) FETCH_SIZE / WRITE_SIZE / MFMA counter passes (separate --pmc runs, no trace domains) for fp32 and
# bf16 storage.  Everything lands in gpurun_out/; copy what is to be judged into profiles/.
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_r04
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r04 -o run -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $root/gpurun_out/bench_prof_r04.log 2>&1; echo "prof rc=$?"
db=$(find $root/gpurun_out/prof_r04 -name "*.db" | head -1)
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/r04_bench_kernel_stats.csv --after-frac 0.3 2> $root/gpurun_out/r04_kernel_stats.txt
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/r04_bench_kernel_grid_stats.csv --after-frac 0.3 --by-grid 2>> $root/gpurun_out/r04_kernel_stats.txt
cat $root/gpurun_out/r04_kernel_stats.txt; rm -rf $root/gpurun_out/prof_r04
cd $root
timeout -k 10 500 python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r04_bench_line_driver_command.json 2>/dev/null; echo "driver-command rc=$?"
timeout -k 10 300 python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04_bench_line_bf16.json 2>/dev/null; echo "bf16 rc=$?"
timeout -k 10 500 python3 bench.py --config 4 --no-cpu-baseline > gpurun_out/r04_bench_line_config4.json 2> gpurun_out/r04_config4.err; echo "config4 rc=$?"
if [ "$1" = "pmc" ]; then
  bash tools/tools_pmc.sh
  bash tools/tools_pmc_bf16.sh
  python3 tools/tools_pmc_summary.py r04
  python3 tools/tools_pmc_summary.py r04 merge-bf16
fi
python3 - <<'PY'
import json
for n in ("r04_bench_line", "r04_bench_line_driver_command", "r04_bench_line_bf16", "r04_bench_line_config4"):
    try:
        d = json.load(open("gpurun_out/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), d.get("conv_fwd_flop_weighted_tflops"))
    except Exception as e:
        print(n, "unreadable:", e)
PY

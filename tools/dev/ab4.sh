#!/bin/bash
# usage (GPU box, repo root): bash tools/dev/ab4.sh -- LDS-resident weights (k_conv_blocks_mfma_wlds) forced onto the 32-plane layers
mkdir -p gpurun_out
run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab4_$tag.json 2> gpurun_out/ab4_$tag.err; }
for rep in 1 2; do
  run f32_base_$rep X=1 --
  run f32_wlds2_$rep AABR_CONV_WLDS=2 --
  run f32_wlds1_$rep AABR_CONV_WLDS=1 --
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab4_*.json")):
    try:
        d = json.load(open(f)); s = d["timing"]["step_ms"]
        print("%-24s %8.2f scenes/s  %7.3f ms  p50 %7.3f  host p50 %7.3f" % (f.split("ab4_")[1][:-5], d["value"], d["ms_per_step"], s["p50"], s["host_enqueue_p50"]))
    except Exception as e:
        print(f, "unreadable", e)
PY

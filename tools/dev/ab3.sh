#!/bin/bash
# usage (GPU box, repo root): bash tools/dev/ab3.sh -- host-side switches of the bf16 step A/B'd on one box, each line twice, interleaved
mkdir -p gpurun_out
run() { # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab3_$tag.json 2> gpurun_out/ab3_$tag.err
}
for rep in 1 2; do
  run bf16_base_$rep X=1 -- --dtype bf16
  run bf16_thread_$rep AABR_BENCH_PREFETCH_THREAD=1 -- --dtype bf16
  run bf16_thread_pipe_$rep AABR_BENCH_PREFETCH_THREAD=1 AABR_PLAN_PIPELINE=48 -- --dtype bf16
  run c4_base_$rep X=1 -- --config 4
  run c4_thread_$rep AABR_BENCH_PREFETCH_THREAD=1 -- --config 4
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab3_*.json")):
    try:
        d = json.load(open(f)); s = d["timing"]["step_ms"]
        print("%-24s %8.2f scenes/s  %7.3f ms  p50 %7.3f  host p50 %7.3f" % (f.split("ab3_")[1][:-5], d["value"], d["ms_per_step"], s["p50"], s["host_enqueue_p50"]))
    except Exception as e:
        print(f, "unreadable", e)
PY

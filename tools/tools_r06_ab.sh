#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_r06_ab.sh [notest] -- round-6 switches A/B'd on ONE box, each line twice, interleaved:
# job-list stream builders (AABR_GEOM_JOBS), bf16 add fusion (AABR_PLAN_FUSE_ADDS_BF16), the launcher thread (AABR_PLAN_PIPELINE)
mkdir -p gpurun_out
if [ "${1:-}" != notest ]; then
  python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06_gputests.txt
fi
run() { # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/r06ab_$tag.json 2> gpurun_out/r06ab_$tag.err
}
for rep in 1 2; do
  run bf16_base_$rep X=1 -- --dtype bf16
  run bf16_nojobs_$rep AABR_GEOM_JOBS=0 -- --dtype bf16
  run bf16_nofuse_$rep AABR_PLAN_FUSE_ADDS_BF16=0 -- --dtype bf16
  run bf16_pipe_$rep AABR_PLAN_PIPELINE=48 -- --dtype bf16
  run f32_base_$rep X=1 --
  run f32_nojobs_$rep AABR_GEOM_JOBS=0 --
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ab_*.json")):
    try:
        d = json.load(open(f)); s = d["timing"]["step_ms"]
        print("%-24s %8.2f scenes/s  %7.3f ms  p50 %7.3f  host p50 %7.3f" % (f.split("r06ab_")[1][:-5], d["value"], d["ms_per_step"], s["p50"], s["host_enqueue_p50"]))
    except Exception as e:
        print(f, "unreadable", e)
PY

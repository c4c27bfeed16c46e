run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/r06ab_$tag.json 2> gpurun_out/r06ab_$tag.err; }
rm -f gpurun_out/r06ab_*.json
for rep in 1 2; do
  run f32_base_$rep X=1 --
  run f32_mainprio_$rep AABR_BENCH_MAIN_PRIORITY=1 --
  run f32_mainprio_nojobs_$rep AABR_BENCH_MAIN_PRIORITY=1 AABR_GEOM_JOBS=0 --
  run bf16_base_$rep X=1 -- --dtype bf16
  run bf16_mainprio_$rep AABR_BENCH_MAIN_PRIORITY=1 -- --dtype bf16
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

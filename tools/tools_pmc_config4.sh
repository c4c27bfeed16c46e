#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc_config4.sh -- FETCH_SIZE and WRITE_SIZE in separate --pmc passes over
# `bench.py --config 4` (one 1.5 M-point scene, bf16); per-kernel averages by tools/tools_pmc_summary.py-style reading
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc4_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc4_$c -- python3 $root/bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0 > $root/gpurun_out/pmc4_$c.log 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "gpurun_out", "pmc4_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            n = r["Kernel_Name"]
            if not any(t in n for t in ("k_conv_narrow", "k_conv_cs", "k_submanifold_table", "k_conv_table", "k_brick_", "k_points_",
                                        "k_voxel_mean", "k_build_tileT", "k_fill_offset_pairs")):
                continue
            key = n.split("(")[0].replace("aabr::", "").replace("void ", "") + "|grid=" + r.get("Grid_Size", "?")
            acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
    out[c] = acc
keys = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]))
with open(os.path.join(root, "gpurun_out", "pmc4_summary.txt"), "w") as fh:
    fh.write("# bench.py --config 4, separate --pmc passes; KB per launch (FETCH_SIZE x 2 per the gfx950 note = bytes fetched)\n")
    for k in keys:
        f = out["FETCH_SIZE"].get(k, [0, 0]); w = out["WRITE_SIZE"].get(k, [0, 0])
        if f[1] < 2:
            continue
        fa, wa = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
        fh.write("%-70s launches %4d  FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  2F+W %8.1f MB\n" % (k, f[1], fa, wa, (2 * fa + wa) / 1024))
print(open(os.path.join(root, "gpurun_out", "pmc4_summary.txt")).read()[:3000])
PY
# KEEP=1: leave the raw passes for `tools_pmc_summary.py <tag> merge-c4`
[ "${KEEP:-0}" = 1 ] || rm -rf $root/gpurun_out/pmc4_FETCH_SIZE $root/gpurun_out/pmc4_WRITE_SIZE

"""Summarise the PMC passes of tools_pmc.sh into profiles/<tag>_pmc_fetch_write_per_kernel.json and
profiles/<tag>_pmc_mfma_conv_kernels.json (tag = argv[1], default r03).  Keys are "<kernel>|grid=<work-items>": one kernel name covers many layer
instances, the grid size tells them apart (bench.py looks its dominant instance up by both)."""
import collections, csv, glob, json, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("aabr::", "").replace(" ", "")


def rows(pat):
    files = glob.glob(pat, recursive=True)
    if not files:
        return []
    return list(csv.DictReader(open(max(files, key=os.path.getmtime))))


def agg(pat, counter=None):
    d = collections.defaultdict(list)
    for r in rows(pat):
        if counter and r.get("Counter_Name") != counter:
            continue
        d["%s|grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    return d


if len(sys.argv) > 2 and sys.argv[2] == "merge-c4":
    # instances of the `--config 4` passes (tools/tools_pmc_config4.sh, run with KEEP=1) that the profile does not have:
    # `bench.py --config 4` looks its dominant instance's traffic up in the same file
    path = "profiles/%s_pmc_fetch_write_per_kernel.json" % TAG
    doc = json.load(open(path))
    fb = agg("gpurun_out/pmc4_FETCH_SIZE/**/*counter_collection.csv")
    wb = agg("gpurun_out/pmc4_WRITE_SIZE/**/*counter_collection.csv")
    added = 0
    for k, v in fb.items():
        if k in doc["kernels"]:
            continue
        wv = wb.get(k, [0.0])
        doc["kernels"][k] = {"launches": len(v), "FETCH_SIZE_KB_avg": round(sum(v) / len(v), 1),
                             "WRITE_SIZE_KB_avg": round(sum(wv) / len(wv), 1), "from": "bench.py --config 4"}
        added += 1
    doc["note"] += "  Entries marked `from: bench.py --config 4` come from the same passes over the configs[4] line."
    json.dump(doc, open(path, "w"), indent=1)
    print("added", added, "config-4 instances")
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "merge-bf16":
    # instances of the `--dtype bf16` passes (tools/tools_pmc_bf16.sh) that the fp32 profile does not have
    path = "profiles/%s_pmc_fetch_write_per_kernel.json" % TAG
    doc = json.load(open(path))
    fb = agg("gpurun_out/pmcbf_FETCH_SIZE/**/*counter_collection.csv")
    wb = agg("gpurun_out/pmcbf_WRITE_SIZE/**/*counter_collection.csv")
    added = 0
    for k, v in fb.items():
        if k in doc["kernels"]:
            continue
        wv = wb.get(k, [0.0])
        doc["kernels"][k] = {"launches": len(v), "FETCH_SIZE_KB_avg": round(sum(v) / len(v), 1),
                             "WRITE_SIZE_KB_avg": round(sum(wv) / len(wv), 1), "from": "bench.py --dtype bf16"}
        added += 1
    doc["note"] += "  Entries marked `from: bench.py --dtype bf16` come from the same passes over the bf16-storage line."
    json.dump(doc, open(path, "w"), indent=1)
    print("added", added, "instances")
    for k, v in sorted(((k, v) for k, v in doc["kernels"].items() if v.get("from")), key=lambda kv: -kv[1]["FETCH_SIZE_KB_avg"] * kv[1]["launches"])[:6]:
        print(k[:80], v)
    sys.exit(0)
f = agg("gpurun_out/pmc_FETCH_SIZE/**/*counter_collection.csv")
w = agg("gpurun_out/pmc_WRITE_SIZE/**/*counter_collection.csv")
out = {}
for k, v in f.items():
    wv = w.get(k, [0.0])
    out[k] = {"launches": len(v), "FETCH_SIZE_KB_avg": round(sum(v) / len(v), 1),
              "WRITE_SIZE_KB_avg": round(sum(wv) / len(wv), 1)}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 6 "
                   "--warmup 2 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0` (tools/tools_pmc.sh); KB per launch, averaged over the "
                   "launches of one (kernel, grid size).  gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x "
                   "(MI355X_MICROARCH.md, HBM section); other access widths are uncalibrated.",
           "kernels": out}, open("profiles/%s_pmc_fetch_write_per_kernel.json" % TAG, "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1]["FETCH_SIZE_KB_avg"] * kv[1]["launches"])[:10]
for k, v in top:
    print(k[:80], v)

names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU_MFMA_MOPS_F32")
m = {n: agg("gpurun_out/pmc_MFMA/**/*counter_collection.csv", n) for n in names}
mo = {}
for k in m["SQ_VALU_MFMA_BUSY_CYCLES"]:
    if "conv" not in k:
        continue
    e = {"launches": len(m["SQ_VALU_MFMA_BUSY_CYCLES"][k])}
    for n in names:
        v = m[n].get(k, [0.0])
        e[n + "_avg"] = round(sum(v) / len(v), 1)
    # MFMA-pipe busy fraction: busy cycles summed over the SIMDs / (4 SIMDs x 256 CUs x elapsed shader cycles);
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    if e["GRBM_GUI_ACTIVE_avg"] > 0:
        e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (4 * 256 * e["GRBM_GUI_ACTIVE_avg"] / 8), 4)
    mo[k] = e
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES "
                   "SQ_INSTS_VALU_MFMA_MOPS_F32 over the same bench command; averages per (kernel, grid size); "
                   "mfma_busy_frac = MFMA busy cycles / (1024 SIMDs x elapsed cycles)", "kernels": mo},
          open("profiles/%s_pmc_mfma_conv_kernels.json" % TAG, "w"), indent=1)
for k, v in sorted(mo.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES_avg"] * kv[1]["launches"])[:10]:
    print(k[:70], v.get("mfma_busy_frac"), v["launches"])

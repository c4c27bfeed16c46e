#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc_lds.sh  -- LDS-pipe counters beside the MFMA-pipe ones for the convolution
# kernels, fp32 and bf16 storage (own --pmc passes, no trace domains): is the bf16 wide kernel LDS-bound?
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in f32 bf16; do
  rm -rf $root/gpurun_out/pmc_LDS_$d
  timeout -k 10 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $root/gpurun_out/pmc_LDS_$d -- python3 $root/bench.py --dtype $d --batches 1 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0 > $root/gpurun_out/pmc_LDS_$d.log 2>&1
  echo "$d rc=$?"
done
cd $root && python3 - <<'PY'
import collections, csv, glob, json, os, re
def short(name):
    name = re.sub(r"^void ", "", name); name = re.sub(r"\(.*\)$", "", name)
    return name.replace("aabr::", "").replace(" ", "")
out = {}
for d in ("f32", "bf16"):
    files = glob.glob("gpurun_out/pmc_LDS_%s/**/*counter_collection.csv" % d, recursive=True)
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = "%s|grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
        if "conv" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # summed over the 8 XCDs
        if gui <= 0:
            continue
        res[k] = dict(launches=len(c["GRBM_GUI_ACTIVE"]), cycles=round(gui, 1),
                      lds_active_frac=round(m.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui * 256), 4),
                      lds_bank_conflict_frac=round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / (gui * 256), 4),
                      mfma_busy_frac=round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 4),
                      lds_insts=round(m.get("SQ_INSTS_LDS", 0.0), 1))
    out[d] = res
    top = sorted(res.items(), key=lambda kv: -kv[1]["cycles"] * kv[1]["launches"])[:8]
    print("==", d)
    for k, v in top:
        print(k[:70], v)
json.dump({"note": "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS over "
                   "`bench.py --dtype {f32,bf16} --batches 1 --steps 4 --warmup 1 ...` (tools/tools_pmc_lds.sh); per (kernel, grid): "
                   "lds_active_frac = SQ_LDS_IDX_ACTIVE / (256 CUs x elapsed cycles) (rocprofv3's LdsUtil), mfma_busy_frac = MFMA busy "
                   "cycles / (1024 SIMDs x elapsed cycles)", "kernels": out}, open("gpurun_out/r06_pmc_lds_conv_kernels.json", "w"), indent=1)
PY

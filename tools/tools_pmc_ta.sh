#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc_ta.sh -- which unit bounds the dominant convolution instance?  Texture-addresser /
# vector-L1 counters beside the MFMA and LDS ones, on the instance ALONE (tools/tools_dominant_instance.py, 30 launches), fp32 and
# bf16 storage; own --pmc passes, no trace domains.
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $root/gpurun_out/pmc_list_avail.txt 2>&1
pass() { # tag dtype counters...
  tag=$1; d=$2; shift 2
  rm -rf $root/gpurun_out/pmc_TA_${tag}_$d
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $root/gpurun_out/pmc_TA_${tag}_$d -- python3 $root/tools/tools_dominant_instance.py $d 30 > $root/gpurun_out/pmc_TA_${tag}_$d.log 2>&1
  echo "$tag $d rc=$?"
}
for d in bf16 f32; do
  # (a TA_TA_BUSY / TA_BUFFER_*_WAVEFRONTS pass did not finish within 300 s on this pool: left out)
  pass stall $d TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE
  pass tcp $d TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_GATE_EN1 TCP_GATE_EN2 GRBM_GUI_ACTIVE
  pass sq $d SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass mfma $d SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
done
cd $root && python3 - <<'PY'
import collections, csv, glob, json, os
out = {}
for d in ("bf16", "f32"):
    m = {}
    for tag in ("stall", "tcp", "sq", "mfma"):
        files = glob.glob("gpurun_out/pmc_TA_%s_%s/**/*counter_collection.csv" % (tag, d), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if "k_conv_cs" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n, v in acc.items():
            v = v[len(v) // 3:]                      # the later launches (caches and clocks settled)
            m[(tag, n)] = sum(v) / len(v)
    res = {}
    for (tag, n), v in m.items():
        gui = m.get((tag, "GRBM_GUI_ACTIVE"), 0.0) / 8.0          # summed over the 8 XCDs
        if n == "GRBM_GUI_ACTIVE" or gui <= 0:
            continue
        res[n] = dict(per_launch=round(v, 1), per_cu_cycle=round(v / (gui * 256), 4), cycles=round(gui, 1))
    out[d] = res
    print("==", d)
    for n, v in sorted(res.items()):
        print("  %-34s %16.1f per launch   %8.4f per CU and elapsed cycle (%.0f cycles)" % (n, v["per_launch"], v["per_cu_cycle"], v["cycles"]))
json.dump({"note": "rocprofv3 --pmc passes over `tools/tools_dominant_instance.py {bf16,f32} 30` (tools/tools_pmc_ta.sh): k_conv_cs on the "
                   "dominant rule book alone; per counter: mean over the later two thirds of the launches, and divided by (256 CUs x "
                   "elapsed cycles = GRBM_GUI_ACTIVE / 8 of the same pass)", "kernels": out},
          open("gpurun_out/r06_pmc_ta_dominant.json", "w"), indent=1)
PY

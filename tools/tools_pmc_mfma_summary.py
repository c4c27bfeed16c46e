"""Summarise tools_pmc_mfma.sh passes into profiles/r01_pmc_mfma_conv_kernels.json"""
import collections, csv, glob, json, os, sys
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 "
               "over tools/tools_conv_bench.py; per-dispatch averages of the forward conv / dW kernels, keyed by kernel "
               "and grid.  GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: "
               "mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).", "runs": {}}
for tag in sys.argv[1:]:
    f = max(glob.glob('gpurun_out/pmc_mfma_%s/*/*counter_collection.csv' % tag), key=os.path.getmtime)
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_conv' not in n:
            continue
        key = n.split('(')[0].replace('void ', '') + " grid=" + r['Grid_Size']
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
    run = {}
    for k, c in d.items():
        e = {cn: round(sum(v) / len(v)) for cn, v in c.items()}
        if e.get('GRBM_GUI_ACTIVE'):
            e['mfma_busy'] = round(e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (e['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        e['dispatches'] = len(next(iter(c.values())))
        run[k] = e
    out["runs"][tag] = run
json.dump(out, open('profiles/r01_pmc_mfma_conv_kernels.json', 'w'), indent=1)
for tag, run in out["runs"].items():
    for k, e in sorted(run.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0))[:8]:
        print(tag, k[:70], e.get('mfma_busy'))

#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_round5.sh  -- the measurements profiles/r05_* come from: the scatter-floor
# microbenchmark, the geometry and kernel A/B tables, kernel-trace summaries of the three bench lines, the PMC passes.
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd $root
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/dev/scatter_floor.hip -o /tmp/scatter_floor && timeout -k 10 120 /tmp/scatter_floor > gpurun_out/r05_scatter_floor.txt 2>&1
echo "floor rc=$?"
timeout -k 10 300 python tools/tools_brick_bench.py > gpurun_out/r05_brick_geometry_ab.txt 2> gpurun_out/r05_brick_geometry_ab.err; echo "geom rc=$?"
# (tools_rb_ab.py and k_conv_rb were removed in round 6; profiles/r05_conv_rb_ab.txt is the record)
(cat profiles/r05_conv_dw_ab.head; timeout -k 10 300 python tools/tools_dw_ab.py bf16 2>/dev/null; timeout -k 10 300 python tools/tools_dw_ab.py f32 2>/dev/null) > gpurun_out/r05_conv_dw_ab.txt; echo "dw rc=$?"
bash tools/tools_profile_cmd.sh r05_fp32 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r05_prof_fp32.txt 2>&1
bash tools/tools_profile_cmd.sh r05_bf16 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 > gpurun_out/r05_prof_bf16.txt 2>&1
bash tools/tools_profile_cmd.sh r05_c4 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config 4 > gpurun_out/r05_prof_c4.txt 2>&1
for d in f32 bf16; do
  cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_dom_$d
  timeout -k 10 300 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_dom_$d -o run -- python3 $root/tools/tools_dominant_instance.py $d 200 > $root/gpurun_out/r05_dominant_instance_$d.txt 2>&1
  db=$(find $root/gpurun_out/prof_dom_$d -name "*.db" | head -1)
  python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/r05_dominant_instance_kernel_stats_$d.csv --by-grid --last 200 2>> $root/gpurun_out/r05_dominant_instance_$d.txt
  rm -rf $root/gpurun_out/prof_dom_$d
  cd $root
done
echo "traces done"
bash tools/tools_pmc.sh > gpurun_out/r05_pmc.log 2>&1
python3 tools/tools_pmc_summary.py r05 > gpurun_out/r05_pmc_summary.txt 2>&1
bash tools/tools_pmc_bf16.sh >> gpurun_out/r05_pmc.log 2>&1
python3 tools/tools_pmc_summary.py r05 merge-bf16 >> gpurun_out/r05_pmc_summary.txt 2>&1
cp profiles/r05_pmc_*.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_MFMA gpurun_out/pmcbf_FETCH_SIZE gpurun_out/pmcbf_WRITE_SIZE
echo "pmc done"
bash tools/tools_pmc_config4.sh > gpurun_out/r05_pmc_config4.txt 2>&1
bash tools/tools_pmc_lds.sh > gpurun_out/r05_pmc_lds.log 2>&1

rm -rf gpurun_out/pmc_LDS_f32 gpurun_out/pmc_LDS_bf16
echo "all done"

#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_round6.sh [part ...]  -- the measurements profiles/r06_* come from.
#   parts: dump (one step launch by launch, fp32 + bf16), stats (kernel-trace summaries of the three bench commands),
#          dom (dominant instance alone, fp32 + bf16), scatter (per-kernel trace of the geometry A/B tool), pmc (FETCH / WRITE /
#          MFMA / LDS passes, fp32 + bf16 + configs[4] merged into one profile), soak (determinism); default: all
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out
parts="${*:-dump stats dom scatter pmc soak}"
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
if has dump; then
  bash tools/tools_step_dump.sh r06_f32 > gpurun_out/r06_dump_f32.log 2>&1
  bash tools/tools_step_dump.sh r06_bf16 --dtype bf16 > gpurun_out/r06_dump_bf16.log 2>&1
  cp gpurun_out/step_dump_r06_f32.txt gpurun_out/r06_step_dump_f32.txt
  cp gpurun_out/step_dump_r06_bf16.txt gpurun_out/r06_step_dump_bf16.txt
  echo "dump done"
fi
if has stats; then
  bash tools/tools_profile_cmd.sh r06_fp32 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_prof_fp32.txt 2>&1
  bash tools/tools_profile_cmd.sh r06_bf16 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 > gpurun_out/r06_prof_bf16.txt 2>&1
  bash tools/tools_profile_cmd.sh r06_c4 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config 4 > gpurun_out/r06_prof_c4.txt 2>&1
  echo "stats done"
fi
if has dom; then
  for d in f32 bf16; do
    cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_dom_$d
    timeout -k 10 300 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_dom_$d -o run -- python3 $root/tools/tools_dominant_instance.py $d 200 > $root/gpurun_out/r06_dominant_instance_$d.txt 2>&1
    db=$(find $root/gpurun_out/prof_dom_$d -name "*.db" | head -1)
    python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/r06_dominant_instance_kernel_stats_$d.csv --by-grid --last 200 2>> $root/gpurun_out/r06_dominant_instance_$d.txt
    rm -rf $root/gpurun_out/prof_dom_$d
    cd $root
  done
  echo "dom done"
fi
if has scatter; then
  timeout -k 10 300 python tools/tools_brick_bench.py > gpurun_out/r06_brick_geometry_ab.txt 2> gpurun_out/r06_brick_geometry_ab.err; echo "geom rc=$?"
  bash tools/tools_profile_cmd.sh r06_scat tools/tools_brick_bench.py > gpurun_out/r06_prof_scat.txt 2>&1
  (echo "# per-(kernel, grid) averages of the scatter's kernels under rocprofv3 --kernel-trace over tools/tools_brick_bench.py";
   echo "# (grid = work-items: 320000 / 1500160 = the point lists of 4 x S80k and of one 1.5 M-point scene; Name,Calls,TotalNs,AverageNs,%,Min,Max)";
   grep -E "k_points|k_brick_mark|k_brick_scan|k_voxel_mean|k_voxel_bin|k_voxel_number|fillBuffer|FillFunctor" gpurun_out/kernel_grid_stats_r06_scat.csv | sort -t, -k3 -n -r | head -60) > gpurun_out/r06_scatter_kernels.txt
  echo "scatter done"
fi
if has pmc; then
  bash tools/tools_pmc.sh > gpurun_out/r06_pmc.log 2>&1
  python3 tools/tools_pmc_summary.py r06 > gpurun_out/r06_pmc_summary.txt 2>&1
  bash tools/tools_pmc_bf16.sh >> gpurun_out/r06_pmc.log 2>&1
  python3 tools/tools_pmc_summary.py r06 merge-bf16 >> gpurun_out/r06_pmc_summary.txt 2>&1
  KEEP=1 bash tools/tools_pmc_config4.sh > gpurun_out/r06_pmc_config4.txt 2>&1
  python3 tools/tools_pmc_summary.py r06 merge-c4 >> gpurun_out/r06_pmc_summary.txt 2>&1
  cp profiles/r06_pmc_*.json gpurun_out/ 2>/dev/null
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_MFMA gpurun_out/pmcbf_FETCH_SIZE gpurun_out/pmcbf_WRITE_SIZE gpurun_out/pmc4_FETCH_SIZE gpurun_out/pmc4_WRITE_SIZE
  bash tools/tools_pmc_lds.sh > gpurun_out/r06_pmc_lds.log 2>&1
  true
  rm -rf gpurun_out/pmc_LDS_f32 gpurun_out/pmc_LDS_bf16
  echo "pmc done"
fi
if has soak; then
  out=gpurun_out/r06_determinism_soak.txt
  (echo "# tools/tools_soak.py 60 {f32,bf16} (round 6 end): the bench step in brick-major site order (job-list builders, bf16 adds in the"
   echo "# write-outs, cast runs), 60 times from fixed seeds in separate processes; then the reference row order; then configs[4] x2") > $out
  for d in f32 f32 f32 bf16 bf16 bf16; do timeout -k 10 200 python tools/tools_soak.py 60 $d 2>/dev/null | grep "^steps" >> $out; done
  echo "# AABR_BENCH_SITE_ORDER=first_seen" >> $out
  for d in f32 f32; do AABR_BENCH_SITE_ORDER=first_seen timeout -k 10 200 python tools/tools_soak.py 60 $d 2>/dev/null | grep "^steps" >> $out; done
  echo "# configs[4]" >> $out
  for d in bf16 bf16; do timeout -k 10 300 python tools/tools_soak.py 40 $d config4 2>/dev/null | grep "^steps" >> $out; done
  echo "soak done"
fi
echo "all done"

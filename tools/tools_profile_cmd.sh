#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_profile_cmd.sh <tag> <python script> [args...]
# rocprofv3 kernel trace of one python tool; per-kernel and per-(kernel, grid) totals through tools/rocpd_stats.py
tag=$1; shift
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
script=$root/$1; shift
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o run -- python3 $script "$@" > $root/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/kernel_stats_$tag.csv 2> $root/gpurun_out/kernel_stats_$tag.txt
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/kernel_grid_stats_$tag.csv --by-grid 2>> $root/gpurun_out/kernel_stats_$tag.txt
cat $root/gpurun_out/kernel_stats_$tag.txt
rm -rf $root/gpurun_out/prof_$tag

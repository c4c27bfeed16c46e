#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_profile_r3.sh <tag> [bench args...]
# rocprofv3 kernel trace of the driver's bench command (per-kernel totals through tools/rocpd_stats.py) and the
# un-profiled line of the same command
tag=$1; shift
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o run -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $root/gpurun_out/bench_prof_$tag.log 2>&1; echo "prof rc=$?"
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/kernel_stats_$tag.csv --after-frac 0.3 2> $root/gpurun_out/kernel_stats_$tag.txt; cat $root/gpurun_out/kernel_stats_$tag.txt
cd $root && timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err; echo "bench rc=$?"

#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_step_dump.sh <tag> [bench.py args...]
# rocprofv3 kernel trace of the bench command, then tools/rocpd_step_dump.py: one step of the main stream launch by launch
set -u
: "${GRAFT_REPO_ROOT:?}"
tag=$1; shift
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_$tag -o run -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $root/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_step_dump.py $db --all-streams > $root/gpurun_out/step_dump_$tag.txt 2>&1
python3 $root/tools/rocpd_stats.py $db $root/gpurun_out/kernel_stats_$tag.csv 2> $root/gpurun_out/kernel_stats_$tag.txt
rm -rf $root/gpurun_out/prof_$tag
head -12 $root/gpurun_out/step_dump_$tag.txt

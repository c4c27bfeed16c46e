#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_timeline_cmd.sh <tag> <python script> [args...]
# rocprofv3 kernel trace of one python tool, then tools/rocpd_timeline.py over the second half of the trace: per-stream
# busy time, union of busy intervals, the gaps on the busiest stream
tag=$1; shift
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
script=$root/$1; shift
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_$tag -o run -- python3 $script "$@" > $root/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_timeline.py $db --after-frac 0.5 --step-marker "k_rpn_label_maps<0>" --per-stream > $root/gpurun_out/timeline_$tag.txt 2>&1
rm -rf $root/gpurun_out/prof_$tag
tail -120 $root/gpurun_out/timeline_$tag.txt

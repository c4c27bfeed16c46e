#!/bin/bash
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_fpn
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_fpn -- python3 $root/tools/tools_fpn_time.py 80000 20 1 > $root/gpurun_out/fpn_prof.log 2>&1; echo "rc=$?"; tail -4 $root/gpurun_out/fpn_prof.log

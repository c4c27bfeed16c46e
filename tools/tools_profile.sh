#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/tools_profile.sh <tag> [bench args...]
# runs the GPU tests, a rocprofv3 kernel-trace of bench.py and an un-profiled bench line
tag=$1; shift
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
timeout -k 10 600 python -m pytest $root/tests -q -m gpu --timeout 300 > $root/gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $root/gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 300 --warmup 200 --no-cpu-baseline "$@" > $root/gpurun_out/bench_prof_$tag.log 2>&1; echo "prof rc=$?"
cd $root && timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_$tag.log 2>gpurun_out/bench_$tag.err; echo "bench rc=$?"; cat gpurun_out/bench_$tag.log

#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc_bf16.sh -- FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs) over
# `bench.py --dtype bf16`; tools/tools_pmc_summary.py r03 merge-bf16 adds the instances that are not in the fp32 profile
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmcbf_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmcbf_$c -- python3 $root/bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0 > $root/gpurun_out/pmcbf_$c.log 2>&1
  echo "$c rc=$?"
done

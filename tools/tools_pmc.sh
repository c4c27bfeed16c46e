#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc.sh   -- FETCH_SIZE and WRITE_SIZE in separate passes over bench.py
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_$c -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $root/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done

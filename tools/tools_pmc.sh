#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc.sh   -- FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes over
# the default bench command (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: not one pass); then
# the MFMA-pipe counters in a third pass.  No trace domains in a --pmc run.
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_$c -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0 > $root/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
rm -rf $root/gpurun_out/pmc_MFMA
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $root/gpurun_out/pmc_MFMA -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-prewarm --min-timed-s 0 > $root/gpurun_out/pmc_MFMA.log 2>&1
echo "MFMA rc=$?"

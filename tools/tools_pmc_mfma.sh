#!/bin/bash
# usage (GPU box, repo root): bash tools/tools_pmc_mfma.sh <npts> <voxel_scale> <tag>
# MFMA-pipe counters of the conv kernels over tools_conv_bench.py (its own pass: --pmc only, no trace domains)
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets it)}"
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/pmc_mfma_$3
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $root/gpurun_out/pmc_mfma_$3 -- python3 $root/tools/tools_conv_bench.py $1 $2 > $root/gpurun_out/pmc_mfma_$3.log 2>&1
echo "rc=$?"

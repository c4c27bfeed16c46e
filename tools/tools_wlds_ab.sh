#!/bin/bash
# dev tool (GPU box): A/B the LDS-resident-weight conv kernel against the streaming kernel
root=$GRAFT_REPO_ROOT
for mode in 0 1 2 4; do
  echo "== AABR_CONV_WLDS=$mode"
  AABR_CONV_WLDS=$mode timeout -k 10 120 python $root/tools/tools_conv_bench.py ${1:-80000} ${2:-20} 2>&1 | grep -E "^ *(32|64|128|256)->" | head -6
done

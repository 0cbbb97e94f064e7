#!/bin/bash
# Round 3, pass b: phase decomposition + SQ counters of the two new forward kernels.
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
bash tools/proj_phases.sh $O
cd /tmp
for K in proj_train attn_fwd; do
  SUB=$([ $K = proj_train ] && echo qkv_proj || echo attn_fwd)
  (cd $GRAFT_REPO_ROOT && bash tools/pmc_kernel.sh $K $SUB $O/pmc_$K)
done

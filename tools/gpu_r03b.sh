#!/bin/bash
# Round 3, pass b: phase decomposition + SQ counters of the two new forward kernels.
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
bash tools/proj_phases.sh $O
bash tools/pmc_kernel.sh proj_train qkv_proj $O/pmc_proj_train
bash tools/pmc_kernel.sh attn_fwd attn_fwd $O/pmc_attn_fwd

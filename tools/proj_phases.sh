#!/bin/bash
# Phase decomposition of the split training forward (one phase switched off at a time; DBG instantiations).  Usage: bash tools/proj_phases.sh OUT
O=${1:-gpurun_out/phases}; mkdir -p $O
{
echo "qkv_proj_kernel<2, DBG>: NR_PROJ_DEBUG bits: 1 no table loads, 2 no MFMAs, 4 no Q/K/V^T stores, 8 no x_save stores, 16 no weight-chunk copies"
for d in 0 32 1 2 4 8 16 12 13 15 31; do echo -n "NR_PROJ_DEBUG=$d  "; NR_PROJ_DEBUG=$d python tools/prof_kernel.py proj_train 2>/dev/null | tail -1; done
echo "attn_fwd_kernel<DBG>: NR_ATTNF_DEBUG bits: 1 no operand loads, 2 no exp / normalisation, 4 no ctx stores"
for d in 0 8 1 2 4 5 6 7; do echo -n "NR_ATTNF_DEBUG=$d  "; NR_ATTNF_DEBUG=$d python tools/prof_kernel.py attn_fwd 2>/dev/null | tail -1; done
} | tee $O/proj_phases.txt

#!/bin/bash
# ninth GPU pass of round 6: what bounds the convolution GEMM -- phase switches of the persistent kernel
export TMPDIR=/tmp
O=gpurun_out/r06i
mkdir -p $O
for K in cgemm_dgrad50; do
  echo -n "one tile per workgroup (k_gemm.h MODE 2): "; NR_CONV_GEMM_PERSIST=0 timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  for D in 0 1 2 3 4 8 12 7 16 18; do
    echo -n "persistent NR_CONVGEMM_DEBUG=$D: "; NR_CONVGEMM_DEBUG=$D timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  done
done | tee $O/cgemm_phases.txt
bash tools/pmc_kernel.sh cgemm_dgrad50 conv_gemm $O/pmc_sq_cgemm_dgrad50 > /dev/null 2>&1
cat $O/pmc_sq_cgemm_dgrad50/summary.txt

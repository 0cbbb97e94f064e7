#!/bin/bash
# sixth GPU pass of round 6: where the whole-sequence pooling forward (k_pool4.h) spends its time -- phase switches + SQ counters
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
for K in pool_fwd_flat50 pool_fwd_flat; do
  echo -n "$K production: "; timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  for D in 64 65 66 68 72 76 80 96 124; do
    echo -n "$K NR_POOL_DEBUG=$D: "; NR_POOL_DEBUG=$D timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  done
done | tee $O/pool4_phases.txt
bash tools/pmc_kernel.sh pool_fwd_flat50 pool4_fwd $O/pmc_sq_pool_fwd_flat50 > /dev/null 2>&1
cat $O/pmc_sq_pool_fwd_flat50/summary.txt

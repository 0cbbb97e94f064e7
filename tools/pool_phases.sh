#!/bin/bash
# Phase decomposition of the register-resident pooling backward (k_pool2.h, DBG instantiation): full kernel vs one phase switched off.
# NR_POOL_DEBUG bits: 1 no ctx loads, 2 no dw / softmax backward, 4 no projection MFMAs, 8 no tanh / dpre / dq arithmetic, 16 no dctx
# product, 32 no global stores.  (Bit 64 = nothing switched off, but the DBG instantiation: its own baseline.)
for K in additive_bwd additive_bwd50; do
  for D in 64 65 66 68 72 80 96 127; do
    echo -n "$K NR_POOL_DEBUG=$D: "; NR_POOL_DEBUG=$D NR_POOL2_S50=2 timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  done
  echo -n "$K production: "; NR_POOL2_S50=2 timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
done

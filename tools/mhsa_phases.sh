#!/bin/bash
# Phase decomposition of mhsa_fwd2 (training form: Q / K / V^T / X saves, both dropouts), DBG instantiation with one phase switched off at a
# time.  NR_MHSA_DEBUG bits: 1 no token gather, 2 no projection MFMAs, 4 no attention, 8 no Q/K/V/X saves, 16 no ctx stores; 64 = nothing
# off (the DBG instantiation's own baseline), 95 = everything off (weight streaming + barriers only).
for D in 64 65 66 68 72 80 95; do
  echo -n "mhsa_train NR_MHSA_DEBUG=$D: "; NR_MHSA_DEBUG=$D timeout 120 python tools/prof_kernel.py mhsa_train 2>/dev/null | tail -1
done
echo -n "mhsa_train production: "; timeout 120 python tools/prof_kernel.py mhsa_train 2>/dev/null | tail -1
echo -n "mhsa_infer production: "; timeout 120 python tools/prof_kernel.py mhsa_infer 2>/dev/null | tail -1

#!/bin/bash
# tenth GPU pass of round 6: the lean persistent convolution GEMM -- stand-alone timing, phase switches, parity, in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r06j
mkdir -p $O
for K in cgemm_dgrad50 cgemm_dgrad20; do
  echo -n "$K one tile per workgroup (k_gemm.h MODE 2): "; NR_CONV_GEMM_PERSIST=0 timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  echo -n "$K persistent: "; timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
done | tee $O/cgemm.txt
for D in 32 33 34 35 36 40 44 39; do
  echo -n "persistent DBG instantiation NR_CONVGEMM_DEBUG=$D: "; NR_CONVGEMM_DEBUG=$D timeout 120 python tools/prof_kernel.py cgemm_dgrad50 2>/dev/null | tail -1
done | tee $O/cgemm_phases.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" --timeout 800 ) > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for CFG in "1 1" "0 0" "1 0"; do
  set -- $CFG
  for M in NAML LSTUR; do
    NR_CONV_GEMM_PERSIST=$1 NR_CONV_FWD_GEMM=$2 timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_${M}_$1$2.json
    python - <<PY
import json
d = json.load(open("$O/line_${M}_$1$2.json"))
kb = d["kernel_breakdown_us_per_step"]
print("persist=$1 fwd_gemm=$2 $M ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "conv3" in k})
PY
  done
done | tee $O/ab.txt

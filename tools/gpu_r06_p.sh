#!/bin/bash
# sixteenth GPU pass of round 6: conv forward with half the sequences per workgroup (two workgroups per CU) -- A/B
export TMPDIR=/tmp
O=gpurun_out/r06p
mkdir -p $O
for H in 0 1 0 1; do
  for K in conv_abs conv_title; do echo -n "NR_CONV_HALF_TILE=$H "; NR_CONV_HALF_TILE=$H timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done
done | tee $O/conv_half.txt
NR_CONV_HALF_TILE=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv_fwd" --timeout 500 2>&1 | tail -2
for H in 0 1; do
  NR_CONV_HALF_TILE=$H timeout 600 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_NAML_$H.json
  python - <<PY
import json
d = json.load(open("$O/line_NAML_$H.json"))
kb = d["kernel_breakdown_us_per_step"]
print("half=$H NAML ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "conv3_fwd" in k})
PY
done | tee $O/lines.txt

#!/bin/bash
# eighth GPU pass of round 6: the convolution as a persistent ring GEMM (csrc/k_convgemm.h), forward and data gradient: parity + A/B inside the steps
export TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" --timeout 800 ) > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for CFG in "1 1" "0 0" "1 0" "0 1"; do
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

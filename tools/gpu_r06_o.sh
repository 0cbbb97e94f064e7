#!/bin/bash
# fifteenth GPU pass of round 6: conv3_kernel with the 672-byte LDS row stride (conflict-free fragment reads under the bank model) -- A/B on one box, parity
export TMPDIR=/tmp
O=gpurun_out/r06o
mkdir -p $O
cat > /tmp/ab_cmd.sh <<'EOS'
for K in conv_abs conv_title; do timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done
EOS
bash tools/ab/run_ab.sh bash /tmp/ab_cmd.sh 2>&1 | tee $O/conv_ab.txt
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py -m gpu -q -k "conv or naml" --timeout 1200 ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
bash tools/pmc_kernel.sh conv_abs conv3_kernel $O/pmc_sq_conv_abs > /dev/null 2>&1
grep "BANK_CONFLICT\|IDX_ACTIVE\|MFMA_BUSY\|GRBM" $O/pmc_sq_conv_abs/summary.txt
for M in NAML LSTUR; do
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_$M.json
  python - <<PY
import json
d = json.load(open("$O/line_$M.json"))
kb = d["kernel_breakdown_us_per_step"]
print("$M ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: kb[k] for k in list(kb)[:8]})
PY
done | tee $O/lines.txt

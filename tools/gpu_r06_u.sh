#!/bin/bash
# twenty-first GPU pass of round 6: strided sequence gradient in the flat pooling backward (LSTUR), the whole GPU suite on the new step, lines
export TMPDIR=/tmp
O=gpurun_out/r06u
mkdir -p $O
for M in LSTUR NAML NRMS; do
timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/err_$M.txt | grep '^{' | tail -1 > $O/line_$M.json
python - <<PY
import json
try:
    d = json.load(open("$O/line_$M.json"))
    kb = d["kernel_breakdown_us_per_step"]
    print("$M ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "additive_bwd" in k or "accum" in k})
except Exception as e:
    print("$M FAILED", e); print(open("$O/err_$M.txt").read()[-1500:])
PY
done
timeout 300 python tools/diag_glue2.py LSTUR small 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -20
( timeout 1500 python -m pytest tests -m gpu -x -q --timeout 1200 ) > $O/pytest_gpu_full.txt 2>&1
tail -5 $O/pytest_gpu_full.txt

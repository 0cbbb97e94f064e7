#!/bin/bash
# seventeenth GPU pass of round 6: element tables with eight lanes per output -- parity, NAML line
export TMPDIR=/tmp
O=gpurun_out/r06q
mkdir -p $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py -m gpu -q -k "element or naml or table" --timeout 1200 ) > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 600 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_NAML.json
python - <<PY
import json
d = json.load(open("$O/line_NAML.json"))
kb = d["kernel_breakdown_us_per_step"]
print("NAML ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "element" in k or "conv3_fwd" in k})
PY

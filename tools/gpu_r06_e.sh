#!/bin/bash
# fifth GPU pass of round 6: scatter holding prefetched rows in memory format, pooling forward with the next group's rows requested inside the
# weighted-sum phase; kernel tests; per-model bench lines
export TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
python tools/scatter_ab.py 2>/dev/null | tail -1 | tee $O/scatter_ab_rowquad.txt
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py -m gpu -q --timeout 1200 ) > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for M in NRMS NAML LSTUR; do
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_$M.json
  python - <<PY
import json
d = json.load(open("$O/line_$M.json"))
kb = d["kernel_breakdown_us_per_step"]
print("$M ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: kb[k] for k in list(kb)[:8]})
PY
done | tee $O/lines.txt

#!/bin/bash
# eighteenth GPU pass of round 6: NAML's view level (S = 4) through the flat pooling backward -- parity, NAML line
export TMPDIR=/tmp
O=gpurun_out/r06r
mkdir -p $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py tests/test_zz_bench_scale_gpu.py -m gpu -q -k "pool_flat or naml or NAML" --timeout 1200 ) > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for F in 1 0; do
NR_POOL_FLAT_VIEWS=$F timeout 600 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_NAML_$F.json
python - <<PY
import json
d = json.load(open("$O/line_NAML_$F.json"))
kb = d["kernel_breakdown_us_per_step"]
print("views flat=$F NAML ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "views" in k or "element" in k})
PY
done

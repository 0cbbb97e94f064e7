#!/bin/bash
# twelfth GPU pass of round 6: everything on the guarded wide stores + the persistent data-gradient GEMM: kernel / model tests, bench lines
export TMPDIR=/tmp
O=gpurun_out/r06l
mkdir -p $O
( time timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_gpu.py tests/test_proj_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_model_gpu.py tests/test_zz_bench_scale_gpu.py -m gpu -q --timeout 1500 ) > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for M in NRMS NAML LSTUR; do
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_$M.json
  python - <<PY
import json
d = json.load(open("$O/line_$M.json"))
kb = d["kernel_breakdown_us_per_step"]
print("$M ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: kb[k] for k in list(kb)[:10]})
PY
done | tee $O/lines.txt

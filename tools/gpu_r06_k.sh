#!/bin/bash
# eleventh GPU pass of round 6: the persistent data-gradient GEMM after the fix at the tile boundary: parity (incl. bench scale), timing, in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r06k
mkdir -p $O
( timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_zz_bench_scale_gpu.py -m gpu -q -k "conv" --timeout 1000 ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for K in cgemm_dgrad50 cgemm_dgrad20; do
  echo -n "$K one tile per workgroup (k_gemm.h MODE 2): "; NR_CONV_GEMM_PERSIST=0 timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
  echo -n "$K persistent: "; timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1
done | tee $O/cgemm.txt
for P in 1 0; do
  for M in NAML LSTUR; do
    NR_CONV_GEMM_PERSIST=$P timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_${M}_$P.json
    python - <<PY
import json
d = json.load(open("$O/line_${M}_$P.json"))
kb = d["kernel_breakdown_us_per_step"]
print("persist=$P $M ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "conv3" in k})
PY
  done
done | tee $O/ab.txt
( timeout 1500 python -m pytest tests/test_naml_gpu.py tests/test_lstur_gpu.py -m gpu -q --timeout 1200 ) > $O/pytest_models.txt 2>&1
tail -4 $O/pytest_models.txt

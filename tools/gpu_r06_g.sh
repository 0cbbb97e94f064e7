#!/bin/bash
# seventh GPU pass of round 6: pooling forward after the rework (shift-free tanh, packed fp32, pipelined fragment reads, branch-free stores)
export TMPDIR=/tmp
O=gpurun_out/r06g
mkdir -p $O
for K in pool_fwd_flat50 pool_fwd_flat; do timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done | tee $O/pool4.txt
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "additive or pool or whole" --timeout 800 ) > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for M in NAML LSTUR; do
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/line_$M.json
  python - <<PY
import json
d = json.load(open("$O/line_$M.json"))
kb = d["kernel_breakdown_us_per_step"]
print("$M ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: v for k, v in kb.items() if "additive_fwd" in k})
PY
done | tee $O/lines.txt

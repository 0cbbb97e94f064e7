#!/bin/bash
# Round 3, pass p: pool2_bwd's fused activation gradient takes its relu / dropout mask from the wave's own fragments (bit masks + one
# cross-lane fetch) instead of 8-byte gathers of the activation -- parity tests and the NAML / LSTUR steps.
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_conv_grad_unquantised_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in kb.items() if 'additive' in k})"; }
for m in NAML LSTUR NAML LSTUR; do
  timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_${m}.json" | line "$m"
done

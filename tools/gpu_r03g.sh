#!/bin/bash
# Round 3, pass g: hand-written dX GEMM vs hipBLASLt (kernel and step A/B), conv-gradient tests.
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_conv_grad_unquantised_gpu.py -x -q -s -m gpu -k "dx_gemm or conv_weight" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|Error" $O/pytest_new.log | cut -c1-600 | tail -6
KB_ONLY=dX timeout 200 python tools/kbench_proj.py > $O/kbench_dx.log 2>&1; tail -5 $O/kbench_dx.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in kb.items() if 'dX' in k or 'dx' in k})"; }
for v in "NR_DX_GEMM=0" "NR_DX_GEMM=1" "NR_DX_GEMM=0" "NR_DX_GEMM=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
NR_DX_GEMM=1 timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu > $O/pytest_model_dx1.log 2>&1; echo "model tests with NR_DX_GEMM=1 rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_model_dx1.log

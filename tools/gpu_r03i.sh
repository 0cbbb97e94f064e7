#!/bin/bash
# Round 3, pass i: fragment-read pipelining in the GEMM kernels, 16-byte write-out in qkv_proj, hand-written weight gradients for the conv encoders.
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_conv_grad_unquantised_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|Error" $O/pytest_new.log | cut -c1-600 | tail -6
timeout 400 python tools/kbench_proj.py > $O/kbench.log 2>&1; grep -E "us$|diff" $O/kbench.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in list(kb.items())[:$2]})"; }
for v in "NR_WGRAD_GEMM=1" "NR_WGRAD_GEMM=0" "NR_WGRAD_GEMM=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_NRMS_$(echo $v | tr ' =' '__').json" | line "NRMS $v" 12
done
for M in NAML LSTUR; do
  for v in "NR_WGRAD_GEMM_CONV=1" "NR_WGRAD_GEMM_CONV=0"; do
    env $v timeout 400 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench_$M.err | tee "$O/bench_${M}_$(echo $v | tr ' =' '__').json" | line "$M $v" 14
  done
done

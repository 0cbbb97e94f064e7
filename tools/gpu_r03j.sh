#!/bin/bash
# Round 3, pass j: 4 vs 8 waves per workgroup in the hand-written GEMM kernels (kernel and step A/B).
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
for W in 4 8; do
  NR_GEMM_WAVES=$W timeout 600 python -m pytest tests/test_proj_gpu.py -x -q -m gpu -k "dx_gemm or tn_gemm" > $O/pytest_w$W.log 2>&1; echo "tests waves=$W rc=$?" | tee -a $O/summary.txt
  NR_GEMM_WAVES=$W KB_ONLY=d timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | sed "s/^/waves=$W /"
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in kb.items() if 'dx' in k or 'dW' in k or 'unpack' in k or 'tn_' in k})"; }
for v in "NR_GEMM_WAVES=4 NR_WGRAD_GEMM=0" "NR_GEMM_WAVES=8 NR_WGRAD_GEMM=0" "NR_GEMM_WAVES=8 NR_WGRAD_GEMM=1" "NR_GEMM_WAVES=4 NR_WGRAD_GEMM=0" "NR_GEMM_WAVES=8 NR_WGRAD_GEMM=0"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -m gpu > $O/pytest_graph.log 2>&1; echo "graph tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_graph.log

#!/bin/bash
# Round 3, pass k: pooled attention forward (nr_attn_pool_fwd) and the 2 x 2 wave blocks of the hand-written GEMMs (kernel and step A/B).
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log
for v in "NR_DX_BLOCKS=1 NR_TN_BLOCKS=1" "NR_DX_BLOCKS=0 NR_TN_BLOCKS=0"; do
  env $v timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | grep -E "attn_|additive|dX|dW" | sed "s/^/$v /"
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in kb.items() if 'dx' in k or 'dW' in k or 'tn_' in k or 'attn_' in k or 'additive_fwd' in k})"; }
for v in "NR_X=1" "NR_FWD_POOL=0" "NR_DX_BLOCKS=0" "NR_WGRAD_GEMM=1" "NR_WGRAD_GEMM=1 NR_WGRAD_GEMM_CONV=1" "NR_X=2" "NR_FWD_POOL=0 NR_DX_BLOCKS=0"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/pytest_graph.log 2>&1; echo "graph/train tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_graph.log

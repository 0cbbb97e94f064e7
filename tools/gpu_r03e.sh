#!/bin/bash
# Round 3, pass e: attn_bwd TILE v2, HIP graph of the step, RCCL world-1, step A/B.
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_proj_gpu.py -x -q -m gpu -k "attn_bwd or proj" > $O/pytest_kern.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_kern.log
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -m gpu > $O/pytest_graph.log 2>&1; echo "graph tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/pytest_graph.log
timeout 900 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu > $O/pytest_rccl.log 2>&1; echo "rccl tests rc=$?" | tee -a $O/summary.txt; tail -8 $O/pytest_rccl.log
KB_ONLY=attn_bwd timeout 200 python tools/kbench_proj.py > $O/kbench.log 2>&1; tail -4 $O/kbench.log
{ echo "attn_bwd_kernel<20,4,DBG,TILE> (dctx through LDS): NR_ATTNB_DEBUG bits: 1 no global loads, 4 no dqkv stores"
for d in 0 8 1 4 5; do echo -n "NR_ATTNB_DEBUG=$d  "; NR_ATTNB_DEBUG=$d python tools/prof_kernel.py attn_bwd_hm 2>/dev/null | tail -1; done; } | tee $O/attn_bwd_phases.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), 'eager', d.get('ms_per_step_eager'), 'enq', round(d['host_enqueue_ms_per_step'],3))"; }
for v in "NR_X=1" "NR_ATTN_TILE=0" "NR_X=2"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --no-graph 2>$O/bench_nograph.err | tee $O/bench_nograph.json | line nograph
timeout 300 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench_naml.err | tee $O/bench_naml.json | line NAML_graph
timeout 300 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --no-graph 2>/dev/null | tee $O/bench_naml_nograph.json | line NAML_nograph
tail -5 $O/bench.err

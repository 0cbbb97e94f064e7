#!/bin/bash
# Round 3, pass d: proj (3 WGs/CU, bias in the epilogue), attn_fwd (2 waves per title), attn_bwd TILE form: parity, micro-benchmarks, phases, step A/B.
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py tests/test_rccl_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
timeout 300 python tools/kbench_proj.py > $O/kbench.log 2>&1; tail -12 $O/kbench.log
NR_ATTN_TILE=0 KB_ONLY=attn_bwd timeout 200 python tools/kbench_proj.py > $O/kbench_tile0.log 2>&1; tail -4 $O/kbench_tile0.log
bash tools/proj_phases.sh $O > /dev/null 2>&1; cat $O/proj_phases.txt
{ echo "attn_bwd_kernel<20,5,DBG,TILE>: NR_ATTNB_DEBUG bits: 1 no global loads, 4 no dqkv stores"
for d in 0 8 1 4 5; do echo -n "NR_ATTNB_DEBUG=$d  "; NR_ATTNB_DEBUG=$d python tools/prof_kernel.py attn_bwd_hm 2>/dev/null | tail -1; done
echo "NR_ATTN_TILE=0:"
for d in 0 8 1 4 5; do echo -n "NR_ATTNB_DEBUG=$d  "; NR_ATTN_TILE=0 NR_ATTNB_DEBUG=$d python tools/prof_kernel.py attn_bwd_hm 2>/dev/null | tail -1; done; } | tee $O/attn_bwd_phases.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for v in "NR_FWD_SPLIT=1" "NR_ATTN_TILE=0" "NR_FWD_SPLIT=0 NR_ATTN_TILE=0" "NR_FWD_SPLIT=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log

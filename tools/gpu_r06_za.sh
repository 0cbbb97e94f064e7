#!/bin/bash
# twenty-seventh GPU pass of round 6: the query-vector partial sums inside the accumulate launch (ops.PartSum): tests, A/B lines
export TMPDIR=/tmp
O=gpurun_out/r06za
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "accum" --timeout 600 ) > $O/pytest_kernels.txt 2>&1; tail -2 $O/pytest_kernels.txt
( timeout 1500 python -m pytest tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_graph_gpu.py tests/test_step_fused_gpu.py tests/test_trajectory_gpu.py tests/test_train_fast.py -m gpu -x -q --timeout 1200 ) > $O/pytest_models.txt 2>&1; tail -3 $O/pytest_models.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],3), 'value', round(d['value']), {k: v for k, v in kb.items() if any(t in k for t in ('sum_parts', 'accum'))})"; }
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for M in NAML LSTUR; do for F in 1 0 1 0; do
  NR_PART_SUM=$F timeout 600 python bench.py --model $M $B 2>/dev/null | grep '^{' | tail -1 | ms "$M part_sum=$F" | tee -a $O/ab_part_sum.txt
done; done

#!/bin/bash
# twenty-sixth GPU pass of round 6: (1) tests of the packing / conversion kernels and the LSTUR / NAML models; (2) split-K partitions of the GRU weight
# gradients (NR_TN_P_MANY_TILES); (3) THE before/after of this session on ONE box: the tree of the round's first session (tools/ab/old_tree =
# commit bd2577d, its own library) against this tree, bench.py --steps 20, A B A B per model
export TMPDIR=/tmp
O=gpurun_out/r06z
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "rows_to or gru or accum or score" --timeout 600 ) > $O/pytest_kernels.txt 2>&1; tail -2 $O/pytest_kernels.txt
( timeout 1500 python -m pytest tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_graph_gpu.py tests/test_step_fused_gpu.py tests/test_optim_gpu.py -m gpu -x -q --timeout 1200 ) > $O/pytest_models.txt 2>&1; tail -3 $O/pytest_models.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],3), 'value', round(d['value']), {k: v for k, v in kb.items() if any(t in k for t in ('gru_dW', 'sum_parts', 'pack_gru', 'rows_to_bf16', 'sort_ids', 'scatter_sorted_f32'))})"; }
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for P in 8 16 8 16 24; do
  NR_TN_P_MANY_TILES=$P timeout 600 python bench.py --model LSTUR $B 2>/dev/null | grep '^{' | tail -1 | ms "LSTUR P=$P" | tee -a $O/ab_tn_parts.txt
done
for M in NRMS NAML LSTUR; do
  for r in 1 2; do
    ( cd tools/ab/old_tree && timeout 600 python bench.py --model $M $B 2>/dev/null | grep '^{' | tail -1 ) | ms "$M OLD(bd2577d)" | tee -a $O/ab_session.txt
    timeout 600 python bench.py --model $M $B 2>/dev/null | grep '^{' | tail -1 | ms "$M NEW" | tee -a $O/ab_session.txt
  done
done
( cd tools/ab/old_tree && timeout 600 python bench.py --model LSTUR --shape large $B 2>/dev/null | grep '^{' | tail -1 ) | ms "LSTUR-large OLD(bd2577d)" | tee -a $O/ab_session.txt
timeout 600 python bench.py --model LSTUR --shape large $B 2>/dev/null | grep '^{' | tail -1 | ms "LSTUR-large NEW" | tee -a $O/ab_session.txt

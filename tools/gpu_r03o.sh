#!/bin/bash
# Round 3, pass o: LSTUR evaluation phase B through row indices (nr_gru_fwd_seq_rows) -- parity tests and score_eval A/B.
export TMPDIR=/tmp
O=gpurun_out/r03o
mkdir -p $O
timeout 900 python -m pytest tests/test_lstur_gpu.py tests/test_evaluate_fast.py tests/test_kernels_gpu.py -x -q -m gpu -k "rows or evaluate or gru or LSTUR" > $O/pytest.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log
for v in "NR_EVAL_ROWS=1" "NR_EVAL_ROWS=0"; do
  for shape in small large; do
  env $v timeout 600 python bench.py --model LSTUR --shape $shape --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>$O/bench.err | tee "$O/bench_LSTUR_${shape}_$(echo $v | tr ' =' '__').json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $shape', round(d['value']), d.get('score_eval'))"
  done
done

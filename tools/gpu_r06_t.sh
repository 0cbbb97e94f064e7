#!/bin/bash
# twentieth GPU pass of round 6: the step without framework kernels -- new kernels' parity, model tests, what is left of the glue, A/B lines
export TMPDIR=/tmp
O=gpurun_out/r06t
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "score or accum or rows_to_f32" --timeout 600 ) > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
( timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_graph_gpu.py tests/test_optim_gpu.py tests/test_train_fast.py tests/test_trajectory_gpu.py -m gpu -x -q --timeout 1200 ) > $O/pytest_models.txt 2>&1
tail -15 $O/pytest_models.txt
for M in NRMS NAML LSTUR; do
  timeout 300 python tools/diag_glue2.py $M small > $O/glue2_$M.txt 2>&1
  echo "== $M"; grep -v "Warning\|amdgpu.ids" $O/glue2_$M.txt | tail -30
done
for M in NRMS NAML LSTUR; do
for F in 1 0; do
NR_BENCH_CRITERION=$F timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/err_${M}_$F.txt | grep '^{' | tail -1 > $O/line_${M}_$F.json
python - <<PY
import json
try:
    d = json.load(open("$O/line_${M}_$F.json"))
    print("$M criterion=$F ms", round(d["ms_per_step"], 3), "loss", d.get("loss"))
except Exception as e:
    print("$M criterion=$F FAILED", e); print(open("$O/err_${M}_$F.txt").read()[-1500:])
PY
done
done

#!/bin/bash
# twenty-fourth GPU pass of round 6: step buffers (token store / GRU history reused), adjacent text streams; tests, glue, lines
export TMPDIR=/tmp
O=gpurun_out/r06x
mkdir -p $O
( timeout 1500 python -m pytest tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_graph_gpu.py tests/test_train_fast.py tests/test_step_fused_gpu.py tests/test_trajectory_gpu.py tests/test_config_knobs_gpu.py -m gpu -x -q --timeout 1200 ) > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for M in NAML LSTUR; do
  timeout 300 python tools/diag_glue2.py $M small 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -20
done
for M in NAML LSTUR NRMS; do
timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/err_$M.txt | grep '^{' | tail -1 > $O/line_$M.json
python - <<PY
import json
try:
    d = json.load(open("$O/line_$M.json"))
    print("$M ms", round(d["ms_per_step"], 3))
except Exception as e:
    print("$M FAILED", e); print(open("$O/err_$M.txt").read()[-1500:])
PY
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3

#!/bin/bash
# first GPU pass of round 6: the new tests, a default bench line, the launcher record (called through tools/run_r06_launcher_on_gpu.sh)
export TMPDIR=/tmp
O=gpurun_out/r06a
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_train_fast.py tests/test_trajectory_gpu.py tests/test_training_parity_gpu.py tests/test_optim_gpu.py tests/test_evaluate_fast.py \
    -m gpu -q -x --timeout 900 ) > $O/pytest_new.txt 2>&1
tail -15 $O/pytest_new.txt
cp gpurun_out/trajectory_*.json gpurun_out/train_parity_fixture_*.json $O/ 2>/dev/null
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.txt 2>&1
grep '^{' $O/bench_default.txt | tail -1 > $O/bench_line_NRMS_small.json
python - <<PY
import json
d = json.load(open("$O/bench_line_NRMS_small.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print(json.dumps(d["roofline"].get("top3"), indent=1)[:3000])
print(json.dumps(d.get("score_eval", {}).get("roofline"), indent=1))
print(json.dumps(d.get("score_eval", {}).get("metrics_fp32_vs_f64"), indent=1))
print({k: (v["value"], v["ms_per_step"]) for k, v in d.get("other_workloads", {}).items()})
print(json.dumps({k: d["train_parity"].get(k) for k in ("diff_auc", "stderr_diff_auc", "z_auc", "diff_ndcg10", "z_ndcg10", "engine_auc", "reference_auc", "error")}, indent=1))
print(json.dumps(d["cpu_baseline"], indent=1)[:1500])
PY

#!/bin/bash
# fourth GPU pass of round 6: scatter with the workgroup merge, the whole-sequence pooling forward (A/B), tests, 32-seed parity
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
for SP in 64 128 256; do NR_SCATTER_SPAN=$SP python tools/scatter_ab.py 2>/dev/null | tail -1; done | tee $O/scatter_ab_merge.txt
for F in 0 1; do
  NR_POOL_FWD_FLAT=$F timeout 600 python bench.py --model NAML --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/naml_poolfwd_$F.json
  python - <<PY
import json
d = json.load(open("$O/naml_poolfwd_$F.json"))
kb = d["kernel_breakdown_us_per_step"]
print("NR_POOL_FWD_FLAT=$F NAML ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "additive_fwd" in k or "scatter" in k})
PY
done | tee $O/naml_poolfwd_ab.txt
( time timeout 2700 python -m pytest tests/test_kernels_gpu.py tests/test_rccl_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_training_parity_gpu.py \
    tests/test_zz_bench_scale_gpu.py tests/test_generic_gpu.py tests/test_evaluate_fast.py -m gpu -q --timeout 1500 ) > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
cp gpurun_out/train_parity_fixture_*.json $O/ 2>/dev/null
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-parity ) > $O/bench_default.txt 2>&1
grep '^{' $O/bench_default.txt | tail -1 > $O/bench_line_NRMS_small.json
python - <<PY
import json
d = json.load(open("$O/bench_line_NRMS_small.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print({k: (round(v["value"]), round(v["ms_per_step"], 3)) for k, v in d.get("other_workloads", {}).items()})
kb = d["kernel_breakdown_us_per_step"]; print({k: kb[k] for k in list(kb)[:9]})
for k, v in d.get("other_workloads", {}).items():
    print(k, json.dumps(v["kernel_breakdown_us_per_step"]))
for m in ("NRMS", "NAML", "LSTUR"):
    try:
        t = json.load(open(f"$O/train_parity_fixture_{m}.json"))
        print(m, {k: round(t[k], 5) for k in ("mean_engine_auc", "mean_reference_auc", "diff_auc", "stderr_diff_auc", "z_auc", "z_ndcg10")}, t["engine_below_reference_pairs"], t["pairs"])
    except Exception as e:
        print(m, e)
PY

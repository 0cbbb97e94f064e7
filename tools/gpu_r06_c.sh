#!/bin/bash
# third GPU pass of round 6: general-geometry tests again, DP segments over RCCL, NAML / LSTUR after the two-phase backward, scatter A/B, bench
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
for SP in 32 64 128 256 512; do NR_SCATTER_SPAN=$SP python tools/scatter_ab.py 2>/dev/null | tail -1; done | tee $O/scatter_ab.txt
( time timeout 2400 python -m pytest tests/test_generic_gpu.py tests/test_rccl_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_kernels_gpu.py \
    tests/test_graph_gpu.py tests/test_zz_bench_scale_gpu.py tests/test_train_fast.py tests/test_optim_gpu.py -m gpu -q --timeout 1200 ) > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_default.txt 2>&1
grep '^{' $O/bench_default.txt | tail -1 > $O/bench_line_NRMS_small.json
python - <<PY
import json
d = json.load(open("$O/bench_line_NRMS_small.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print([(t["kernel"], round(t["avg_us"]), round(t.get("frac_mfma", 0), 3), round(t.get("frac_hbm", 0), 3)) for t in d["roofline"]["top3"]])
print({k: (round(v["value"]), round(v["ms_per_step"], 3)) for k, v in d.get("other_workloads", {}).items()})
print(json.dumps(d["kernel_breakdown_us_per_step"]))
for k, v in d.get("other_workloads", {}).items():
    print(k, json.dumps(v["kernel_breakdown_us_per_step"]))
tp = d.get("train_parity", {})
print({k: tp.get(k) for k in ("diff_auc", "stderr_diff_auc", "z_auc", "z_ndcg10")})
PY

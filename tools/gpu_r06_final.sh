#!/bin/bash
# Round-6 final bench lines (after the profile pass: profiles/traffic.json, mfma_busy.json, step_traffic.json are those of THIS library).
export TMPDIR=/tmp
TAG=${1:-r06final}
O=gpurun_out/$TAG
mkdir -p $O
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
r = d["roofline"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "host", round(d["host_enqueue_ms_per_step"], 2), "|", r["kernel"], r["bound"], round(r["frac"], 3), round(r["avg_us"], 1), "traffic", r.get("traffic"), "step traffic", r.get("traffic_total_per_step"))
if r.get("projection_gemm"): print("   projection_gemm frac", round(r["projection_gemm"]["frac"], 3), round(r["projection_gemm"]["avg_us"], 1), "| mfma_busy", r.get("mfma_busy_frac_by_kernel"))
if "value_dropin" in d: print("   dropin", round(d["value_dropin"]["value"]), round(d["value_dropin"]["ms_per_step"], 2), "| score_eval", round(d["score_eval"]["value"]), "| fwd-only", round(d["score_impressions_per_s_fwd_only"]))
if d.get("parity"):
    p = d["parity"]; ms = p.get("multi_seed_n1000", {})
    print("   parity n1000", p["worst_abs_diff_auc_n1000"], p["worst_abs_diff_ndcg10_n1000"], "n5000", p["worst_abs_diff_auc_n5000"], p["worst_abs_diff_ndcg10_n5000"], "| seeds", p.get("seeds"), "max", ms.get("max_abs_diff_auc"), ms.get("max_abs_diff_ndcg10"), p["within_tolerance"], "| models", d.get("parity_models"))
if d.get("train_parity"):
    t = d["train_parity"]; print("   train_parity", {k: t.get(k) for k in ("mean_engine_auc", "mean_reference_auc", "diff_auc", "stderr_diff_auc", "z_auc", "z_ndcg10", "engine_below_reference_pairs", "pairs", "seconds")})
if r.get("top3"): print("   top3", [(t["kernel"], round(t["avg_us"], 1), round(t.get("frac_mfma", 0), 3), round(t.get("frac_hbm", 0), 3), t.get("traffic")) for t in r["top3"]])
if d.get("score_eval", {}).get("roofline"): print("   score_eval roofline", {k: (round(v["avg_us"], 1), round(v["frac"], 3)) for k, v in d["score_eval"]["roofline"].items()}, d["score_eval"].get("metrics_fp32_vs_f64", {}).get("abs_diff_of_means"))
if d.get("cpu_baseline"): print("   cpu", {k: (v if not isinstance(v, dict) else round(v["value"], 1)) for k, v in d["cpu_baseline"].items() if k != "sample"})
if "gather_roofline" in d:
    g = d["gather_roofline"]; print("   gather hbm", round(g["hbm_point"]["achieved"]), "GB/s", round(g["frac"], 3), "| workload", round(g["workload_point"]["achieved"]), "| traffic", g.get("traffic"))
print("   kernels", dict(list(d["kernel_breakdown_us_per_step"].items())[:14]))
PY
}
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_NRMS_small.json 2> $O/bench_line_NRMS_small.err ) 2>&1 | grep real; q $O/bench_line_NRMS_small.json
# the other workloads: timing + roofline legs only (their parity is the test suite's; the CPU baseline and the parity legs are on the default line)
QUICK="--no-train-parity --no-parity --no-cpu-baseline"
timeout 900 python bench.py --model NAML $QUICK > $O/bench_line_NAML_small.json 2> $O/bench_line_NAML_small.err; q $O/bench_line_NAML_small.json
timeout 900 python bench.py --model LSTUR --shape large $QUICK > $O/bench_line_LSTUR_large.json 2> $O/bench_line_LSTUR_large.err; q $O/bench_line_LSTUR_large.json
timeout 900 python bench.py --model LSTUR $QUICK > $O/bench_line_LSTUR_small.json 2> $O/bench_line_LSTUR_small.err; q $O/bench_line_LSTUR_small.json
timeout 900 python bench.py --shape large $QUICK > $O/bench_line_NRMS_large.json 2> $O/bench_line_NRMS_large.err; q $O/bench_line_NRMS_large.json

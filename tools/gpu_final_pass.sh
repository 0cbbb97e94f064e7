#!/bin/bash
# Round-3 measurement pass: parity tests, smoke, HBM traffic and SQ counters of the dominant kernels (first: the bench lines quote them), bench
# lines (all workloads), rocprofv3 kernel stats, the N > 1 code path of bench.py with two ranks sharing the one GPU (gloo).  Usage: bash tools/gpu_final_pass.sh [TAG]
export TMPDIR=/tmp
TAG=${1:-r03final2}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
r = d["roofline"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "host", round(d["host_enqueue_ms_per_step"], 2), "|", r["kernel"], r["bound"], round(r["frac"], 3), round(r["avg_us"], 1), "traffic", r.get("traffic"), "mfma_busy", r.get("mfma_busy_frac_by_kernel"))
if r.get("projection_gemm"): print("   projection_gemm frac", round(r["projection_gemm"]["frac"], 3), round(r["projection_gemm"]["avg_us"], 1))
if "value_dropin" in d: print("   dropin", round(d["value_dropin"]["value"]), round(d["value_dropin"]["ms_per_step"], 2), "| score_eval", round(d["score_eval"]["value"]), "| fwd-only", round(d["score_impressions_per_s_fwd_only"]))
if d.get("parity"):
    p = d["parity"]; print("   parity worst n1000", p["worst_abs_diff_auc_n1000"], p["worst_abs_diff_ndcg10_n1000"], "n5000", p["worst_abs_diff_auc_n5000"], p["worst_abs_diff_ndcg10_n5000"], p["within_tolerance"], "| parity_models", d.get("parity_models"))
if d.get("cpu_baseline"): print("   cpu", {k: (v if not isinstance(v, dict) else round(v["value"], 1)) for k, v in d["cpu_baseline"].items() if k != "sample"})
if "gather_roofline" in d:
    g = d["gather_roofline"]; print("   gather hbm", round(g["hbm_point"]["achieved"]), "GB/s", round(g["frac"], 3), "| workload", round(g["workload_point"]["achieved"]))
print("   kernels", dict(list(d["kernel_breakdown_us_per_step"].items())[:12]))
PY
}
for K in proj_train attn_pool_fwd attn_bwd_hm additive_bwd; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${K}_fetch -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${K}_write -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_write.log 2>&1
done
python tools/pmc_traffic.py $O profiles/r03_pmc_traffic.txt | tee $O/pmc_traffic.txt
rm -rf $O/pmc_*_fetch $O/pmc_*_write
for K in "proj_train qkv_proj" "attn_pool_fwd attn_fwd_kernel" "attn_bwd_hm attn_bwd" "additive_bwd pool2_bwd" "additive_bwd50 pool2_bwd" "conv_abs conv3" "dx_gemm dx_gemm" "tn_gemm tn_gemm"; do
  set -- $K
  bash tools/pmc_kernel.sh $1 $2 $O/pmc_sq_$1 > /dev/null 2>&1
done
python tools/pmc_sq.py $O r03 | tee $O/pmc_sq_summary.txt
# the bench lines below report traffic / mfma_busy only from files measured on THIS library (source hash): take the ones just made
cp $O/traffic.json profiles/traffic.json; cp $O/mfma_busy.json profiles/mfma_busy.json
timeout 900 python bench.py > $O/bench_line_NRMS_small.json 2> $O/bench_line_NRMS_small.err; q $O/bench_line_NRMS_small.json
timeout 900 python bench.py --shape large > $O/bench_line_NRMS_large.json 2> $O/bench_line_NRMS_large.err; q $O/bench_line_NRMS_large.json
timeout 1200 python bench.py --model LSTUR --shape large > $O/bench_line_LSTUR_large.json 2> $O/bench_line_LSTUR_large.err; q $O/bench_line_LSTUR_large.json
timeout 900 python bench.py --model NAML > $O/bench_line_NAML_small.json 2> $O/bench_line_NAML_small.err; q $O/bench_line_NAML_small.json
timeout 900 python bench.py --model LSTUR > $O/bench_line_LSTUR_small.json 2> $O/bench_line_LSTUR_small.err; q $O/bench_line_LSTUR_small.json
for W in "NRMS small" "NAML small" "LSTUR large"; do
  set -- $W
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$1_$2 -o bench -- python bench.py --model $1 --shape $2 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras > $O/under_rocprof_$1_$2.log 2>&1
  DB=$(find $O/prof_$1_$2 -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_$1_$2.csv > /dev/null && python tools/rocpd_gaps.py $DB > $O/gaps_$1_$2.txt 2>&1
  rm -rf $O/prof_$1_$2
done
head -14 $O/kernel_stats_NRMS_small.csv | cut -c1-170
bash tools/gpu_two_ranks_one_gpu.sh > $O/two_ranks.log 2>&1; grep "^rc\[" $O/two_ranks.log; cp gpurun_out/two_ranks_NRMS.log gpurun_out/two_ranks_LSTUR.log $O/ 2>/dev/null

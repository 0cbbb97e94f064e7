#!/bin/bash
# Round-2 first GPU pass: parity tests, smoke, bench lines (small default, large shapes), kernel stats of the NRMS step.
export TMPDIR=/tmp
TAG=${1:-r02a}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
summ() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); sys.exit(0)
r = d["roofline"]
print(d["config"]["workload"][:60], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2),
      "|", r["kernel"], round(r["frac"], 3), round(r["avg_us"], 1))
for k in ("value_dropin", "score_eval"):
    if k in d: print("  ", k, round(d[k]["value"]), {a: b for a, b in d[k].items() if a in ("ms_per_step", "seconds")})
if d.get("parity"):
    p = d["parity"]
    print("   parity worst n1000", p["worst_abs_diff_auc_n1000"], p["worst_abs_diff_ndcg10_n1000"], "n5000", p["worst_abs_diff_auc_n5000"], p["worst_abs_diff_ndcg10_n5000"])
if d.get("cpu_baseline"): print("   cpu", {k: (v if not isinstance(v, dict) else round(v["value"], 1)) for k, v in d["cpu_baseline"].items() if k != "sample"})
if "gather_roofline" in d:
    g = d["gather_roofline"]; print("   gather hbm", round(g["hbm_point"]["achieved"]), "GB/s", round(g["frac"], 3), "| workload", round(g["workload_point"]["achieved"]))
print("   kernels", dict(list(d["kernel_breakdown_us_per_step"].items())[:14]))
PY
}
timeout 900 python bench.py > $O/bench_NRMS_small.json 2> $O/bench_NRMS_small.err; summ $O/bench_NRMS_small.json; tail -3 $O/bench_NRMS_small.err
timeout 600 python bench.py --shape large --no-parity --no-cpu-baseline > $O/bench_NRMS_large.json 2> $O/bench_NRMS_large.err; summ $O/bench_NRMS_large.json; tail -3 $O/bench_NRMS_large.err
timeout 900 python bench.py --model LSTUR --shape large --no-parity --no-cpu-baseline > $O/bench_LSTUR_large.json 2> $O/bench_LSTUR_large.err; summ $O/bench_LSTUR_large.json; tail -3 $O/bench_LSTUR_large.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_NRMS -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras > $O/bench_NRMS_under_rocprof.log 2>&1
DB=$(find $O/prof_NRMS -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_NRMS.csv > /dev/null && head -30 $O/kernel_stats_NRMS.csv
rm -rf $O/prof_NRMS

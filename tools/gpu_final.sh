#!/bin/bash
# Round-end GPU pass: all parity tests, smoke(), the three bench lines, rocprofv3 kernel stats per model, PMC traffic of the dominant kernels.
export TMPDIR=/tmp
TAG=${1:-r01f}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for M in NRMS NAML LSTUR; do
  timeout 900 python bench.py --model $M > $O/bench_$M.json 2> $O/bench_$M.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$M.json").read().strip().splitlines()[-1])
print("$M", round(d["value"]), round(d["ms_per_step"],2), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["parity"]["abs_diff_auc"], d["parity"]["abs_diff_ndcg10"], round(d["cpu_baseline"]["value"],1))
PY
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$M -o bench -- python bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_${M}_under_rocprof.log 2>&1
  DB=$(find $O/prof_$M -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_$M.csv > /dev/null
  rm -rf $O/prof_$M
done
for K in mhsa_train attn_bwd; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${K}_fetch -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${K}_write -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_write.log 2>&1
done
python tools/pmc_summary.py $O mhsa_fwd > $O/pmc_mhsa.txt 2>&1; python tools/pmc_summary.py $O attn_bwd > $O/pmc_attn_bwd.txt 2>&1
cat $O/pmc_mhsa.txt $O/pmc_attn_bwd.txt
rm -rf $O/pmc_*_fetch $O/pmc_*_write

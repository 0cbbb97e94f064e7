#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel stats, PMC traffic of the dominant kernel.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh TAG [skip-tests]
export TMPDIR=/tmp
TAG=${1:-r01}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats.csv > /dev/null
# HBM traffic of the dominant kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for K in mhsa_train attn_bwd; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${K}_fetch -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${K}_write -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_write.log 2>&1
done
python tools/pmc_summary.py $O mhsa_fwd > $O/pmc_mhsa.txt 2>&1
python tools/pmc_summary.py $O attn_bwd > $O/pmc_attn_bwd.txt 2>&1
cat $O/pmc_mhsa.txt $O/pmc_attn_bwd.txt
timeout 200 python tools/kbench.py > $O/kbench.log 2>&1; tail -3 $O/kbench.log

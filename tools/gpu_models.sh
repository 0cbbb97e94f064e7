#!/bin/bash
# Bench + rocprofv3 kernel stats for the NAML and LSTUR legs.  Usage: bash tools/gpu_models.sh TAG
export TMPDIR=/tmp
TAG=${1:-r01m}
O=gpurun_out/$TAG
mkdir -p $O
for M in NAML LSTUR; do
  timeout 900 python bench.py --model $M > $O/bench_$M.json 2> $O/bench_$M.err
  tail -c 2500 $O/bench_$M.json; tail -3 $O/bench_$M.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$M -o bench -- python bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_${M}_under_rocprof.log 2>&1
  DB=$(find $O/prof_$M -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_$M.csv > /dev/null
  rm -rf $O/prof_$M
done

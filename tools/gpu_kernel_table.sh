#!/bin/bash
# rocprofv3 kernel statistics + gap analysis of one workload's eager step.  Usage: bash tools/gpu_kernel_table.sh MODEL SHAPE TAG
export TMPDIR=/tmp
M=${1:-NRMS}; SH=${2:-small}; TAG=${3:-ktab}
O=gpurun_out/$TAG
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_${M}_${SH} -o bench -- python bench.py --model $M --shape $SH --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-train-parity --no-other-workloads > $O/under_rocprof_${M}_${SH}.log 2>&1
DB=$(find $O/prof_${M}_${SH} -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_${M}_${SH}.csv > /dev/null && python tools/rocpd_gaps.py $DB > $O/gaps_${M}_${SH}.txt 2>&1
rm -rf $O/prof_${M}_${SH}
head -40 $O/kernel_stats_${M}_${SH}.csv | cut -c1-150
tail -3 $O/under_rocprof_${M}_${SH}.log | cut -c1-300

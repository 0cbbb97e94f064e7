#!/bin/bash
# Quick tuning pass: kernel stats of the NRMS / NAML benches + environment-knob variants.  Usage: bash tools/gpu_quick.sh TAG
export TMPDIR=/tmp
TAG=${1:-q}
O=gpurun_out/$TAG
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for M in NRMS NAML; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$M -o bench -- python bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_${M}_under_rocprof.log 2>&1
  DB=$(find $O/prof_$M -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_$M.csv > /dev/null
  rm -rf $O/prof_$M
  head -12 $O/kernel_stats_$M.csv | cut -c1-110
done
NR_CONV_VARIANT=0 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | line NAML_conv0
NR_CONV_VARIANT=0 python bench.py --model LSTUR --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | line LSTUR_conv0
NR_ADD_VARIANT=1 python bench.py --model NRMS --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | line NRMS_add1
NR_ADD_VARIANT=1 python bench.py --model NAML --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | line NAML_add1

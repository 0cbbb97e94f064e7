#!/bin/bash
# Generic A/B pass on one MI355X box (one gpurun call = one box: numbers of different calls are not comparable, DVFS).
#   bash tools/gpu_ab.sh TAG [-m MODEL] [-s SHAPE] [-k KBENCH_PATTERN] [-t "pytest args"] -- "ENV=.. ENV=.." "ENV=.." ...
# For every environment-variable set: (optional) tools/kbench_proj.py restricted to KBENCH_PATTERN (stand-alone kernel times), then
# `bench.py --steps 20 --warmup 5` without the CPU / parity / extras legs; prints value, ms/step and the kernel breakdown.  The first set is
# run again at the end (drift check).  Every A/B of profiles/r03_ab_switches.txt was taken this way; the switches are documented where they
# are read (ops.py, csrc/nr_engine.hip) and in INTEGRATION.md.
export TMPDIR=/tmp
TAG=$1; shift
MODEL=NRMS; SHAPE=small; KB=""; TESTS=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do
  case $1 in -m) MODEL=$2; shift 2;; -s) SHAPE=$2; shift 2;; -k) KB=$2; shift 2;; -t) TESTS=$2; shift 2;; *) echo "unknown option $1"; exit 2;; esac
done
shift
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log; fi
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 |', round(d['value']), 'impressions/s', round(d['ms_per_step'],3), 'ms |', dict(list(d['kernel_breakdown_us_per_step'].items())[:14]))"; }
run() {
  if [ -n "$KB" ]; then env $1 KB_ONLY=$KB timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | sed "s/^/$1 | /" | tee -a $O/summary.txt; fi
  env $1 timeout 400 python bench.py --model $MODEL --shape $SHAPE --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err \
    | tee "$O/bench_$(echo $1 | tr ' =' '__').json" | line "$1" | tee -a $O/summary.txt
}
for v in "$@"; do run "$v"; done
run "$1"

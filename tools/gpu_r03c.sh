#!/bin/bash
# Round 3, pass c: v2 of the split forward (coalesced gather / staged stores): parity, micro-benchmarks, phases, step A/B.
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
timeout 300 python tools/kbench_proj.py > $O/kbench_k2.log 2>&1; tail -12 $O/kbench_k2.log
NR_PROJ_KSPLIT=1 KB_ONLY=proj timeout 200 python tools/kbench_proj.py > $O/kbench_k1.log 2>&1; tail -5 $O/kbench_k1.log
bash tools/proj_phases.sh $O > /dev/null 2>&1; cat $O/proj_phases.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for sp in 1 0 1; do
  NR_FWD_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench_split$sp.err | tee $O/bench_split$sp.json | line split$sp
done
NR_FWD_SPLIT=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | tee $O/bench_split2_extras.json | line split2_with_extras
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log

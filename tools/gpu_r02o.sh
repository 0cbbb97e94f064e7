#!/bin/bash
# r02o: batch H2D copies on a side stream (value_dropin A/B), then the full GPU suite
export TMPDIR=/tmp
O=gpurun_out/${1:-r02o}
mkdir -p $O
for rep in 1 2; do
for cfg in "NR_COPY_STREAM=0" "NR_COPY_STREAM=1"; do
for M in NRMS LSTUR; do
  env $cfg timeout 300 python bench.py --model $M --no-parity --no-cpu-baseline --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); v=d['value_dropin']; print('$M $cfg value', round(d['value']), 'ms', round(d['ms_per_step'],3), '| dropin', round(v['value']), round(v['ms_per_step'],3))"
done
done
done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log

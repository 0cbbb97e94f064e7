#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02p}
mkdir -p $O
for M in LSTUR NRMS; do
  timeout 300 python bench.py --model $M --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$M value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2))"
done
timeout 300 python tools/diag_dropin2.py LSTUR small > $O/diag_LSTUR.log 2>&1; grep -v "amdgpu.ids" $O/diag_LSTUR.log | head -40

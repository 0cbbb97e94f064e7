#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02l}
mkdir -p $O
timeout 300 python tools/diag_glue.py NRMS small > $O/glue_NRMS.log 2>&1; tail -50 $O/glue_NRMS.log
for rep in ; do
  timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('NRMS value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_step'],2))"
done

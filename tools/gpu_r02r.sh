#!/bin/bash
# r02r: attn_bwd with CL(P) / CL(dS) transposed on the matrix core instead of recomputed
export TMPDIR=/tmp
O=gpurun_out/${1:-r02r}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "attn_bwd or model or knobs or optim" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=d['kernel_breakdown_us_per_step']; print('NRMS value', round(d['value']), 'ms', round(d['ms_per_step'],3), {a: round(b) for a,b in k.items() if 'attn_bwd' in a})"
done

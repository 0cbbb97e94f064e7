#!/bin/bash
# r02w: attn_bwd persistent-grid size A/B (NR_ATTN_BWD_MAX_WGS; default 6144 workgroups of 4 waves)
export TMPDIR=/tmp
O=gpurun_out/r02w; mkdir -p $O
for rep in 1 2; do
for cap in 0 12288 24576 200000; do
  NR_ATTN_BWD_MAX_WGS=$cap timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=d['kernel_breakdown_us_per_step']; print('NR_ATTN_BWD_MAX_WGS=$cap NRMS value', round(d['value']), 'ms', round(d['ms_per_step'],3), {a: round(b) for a,b in k.items() if 'attn_bwd' in a})"
done
done

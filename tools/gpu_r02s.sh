#!/bin/bash
# r02s: attn_bwd A/B on one box, both variants built there: CL(P) / CL(dS) recomputed (NR_ATTN_RECOMPUTE=1) vs transposed on the matrix core (0)
export TMPDIR=/tmp
O=gpurun_out/${1:-r02s}
mkdir -p $O
for rep in 1 2; do
for v in 1 0; do
  NR_EXTRA_FLAGS="-DNR_ATTN_RECOMPUTE=$v" bash news_recommendation_amd/csrc/build.sh > $O/build_$v.log 2>&1 || { echo "build failed v=$v"; tail -5 $O/build_$v.log; continue; }
  timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=d['kernel_breakdown_us_per_step']; print('NR_ATTN_RECOMPUTE=$v NRMS value', round(d['value']), 'ms', round(d['ms_per_step'],3), {a: round(b) for a,b in k.items() if 'attn_bwd' in a})"
done
done

#!/bin/bash
# last check of a round: full GPU suite, smoke, the default bench line (what the driver runs)
export TMPDIR=/tmp
O=gpurun_out/${1:-r02check}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 900 python bench.py > $O/bench_line_NRMS_small.json 2> $O/bench.err; echo "bench.py wall $(( $(date +%s) - T0 )) s"
python - $O/bench_line_NRMS_small.json <<'PY'
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('value', round(d['value']), 'ms', round(d['ms_per_step'], 3), '| roofline', r['kernel'], round(r['frac'], 4), 'traffic', r['traffic'], '| dropin', round(d['value_dropin']['value']), '| score_eval', round(d['score_eval']['value']), '| parity', d['parity']['within_tolerance'], '| cpu', d['cpu_baseline']['kind'], d['cpu_baseline']['value'])
PY

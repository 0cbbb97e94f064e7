#!/bin/bash
# last GPU pass of round 6 on a FRESH box, the driver's own sequence with the final library: smoke(), the default bench line, then the N > 1 code
# path of bench.py (two ranks on the one GPU over gloo, all three models)
export TMPDIR=/tmp
O=gpurun_out/r06zi
mkdir -p $O
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_line_default.json 2> $O/bench_line_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06zi/bench_line_default.json').read().strip().splitlines()[-1])
print('default line: value', round(d['value']), 'ms', round(d['ms_per_step'], 3), 'roofline', d['roofline']['kernel'], d['roofline']['bound'], round(d['roofline']['frac'], 3), 'traffic', d['roofline'].get('traffic'),
      'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], 'other', {k: round(v['ms_per_step'], 2) for k, v in d.get('other_workloads', {}).items()})
PY
bash tools/gpu_two_ranks_one_gpu.sh 2>&1 | grep -E "rc\[|\"value\"" | cut -c1-400

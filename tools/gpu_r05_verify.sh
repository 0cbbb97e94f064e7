#!/bin/bash
# Final verification after the closing pass: LSTUR tests + smoke(), the LSTUR line (GRU sweeps as one profiled entry) and the default line again.
export TMPDIR=/tmp
O=gpurun_out/r05verify; mkdir -p $O
timeout 600 python -m pytest tests/test_lstur_gpu.py tests/test_training_parity_gpu.py -x -q -k "not naml" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --model LSTUR --shape large --no-train-parity --no-parity --no-cpu-baseline > $O/bench_line_LSTUR_large.json 2> $O/lstur.err; tail -c 1200 $O/lstur.err | tail -3
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05verify/bench_line_LSTUR_large.json').read().strip().splitlines()[-1])
print('LSTUR large', round(d['value']), round(d['ms_per_step'], 3), d['roofline'])
print({k: v for k, v in list(d['kernel_breakdown_us_per_step'].items())[:8]})
PY
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_NRMS_small.json 2> $O/nrms.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05verify/bench_line_NRMS_small.json').read().strip().splitlines()[-1])
r = d['roofline']
print('NRMS small', round(d['value']), round(d['ms_per_step'], 3), r['kernel'], round(r['frac'], 3), r.get('traffic'), r.get('traffic_total_per_step'), r.get('mfma_busy_frac'))
print({k: (round(v['value']), round(v['ms_per_step'], 3), v['roofline']['kernel'], round(v['roofline'].get('frac', 0), 3)) for k, v in d['other_workloads'].items()})
PY

#!/bin/bash
# Round 3, pass f: new parity tests, the default bench line, kernel trace, HBM traffic counters of the new kernels.
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_conv_grad_unquantised_gpu.py tests/test_kernels_gpu.py -x -q -s -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|CNN.weight|Error" $O/pytest_new.log | cut -c1-900 | tail -12
timeout 900 python bench.py > $O/bench_line_NRMS_small.json 2> $O/bench_line_NRMS_small.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03f/bench_line_NRMS_small.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms', round(d['ms_per_step'], 3), 'eager', d.get('ms_per_step_eager'), 'host', round(d['host_enqueue_ms_per_step'], 3))
print('roofline', d['roofline'])
print('parity', {k: v for k, v in d['parity'].items() if k.startswith(('worst', 'within'))}, 'parity_models', d.get('parity_models'))
print('dropin', d.get('value_dropin', {}).get('value'), 'score_eval', d.get('score_eval', {}).get('value'), 'cpu', {k: (v if not isinstance(v, dict) else v.get('value')) for k, v in (d.get('cpu_baseline') or {}).items() if k != 'sample'})
PY
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras > $O/under_rocprof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_NRMS_small.csv > /dev/null && python tools/rocpd_gaps.py $DB > $O/gaps_NRMS_small.txt 2>&1
rm -rf $O/prof
head -16 $O/kernel_stats_NRMS_small.csv | cut -c1-160; tail -5 $O/gaps_NRMS_small.txt
for K in proj_train attn_fwd attn_bwd_hm; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${K}_fetch -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${K}_write -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_write.log 2>&1
done
python tools/pmc_traffic.py $O | tee $O/pmc_traffic.txt
rm -rf $O/pmc_*_fetch $O/pmc_*_write

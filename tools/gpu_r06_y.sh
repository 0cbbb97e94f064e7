#!/bin/bash
# twenty-fifth GPU pass of round 6: qkv_proj without scratch (lane ids re-derived after the gather phase) against the build that spills two registers
# (tools/ab/libnr_engine_old.so), one box: parity of the projection kernels, NRMS step time both ways (A B A B), dispatch gaps before qkv_proj
export TMPDIR=/tmp
O=gpurun_out/r06y
mkdir -p $O
( timeout 600 python -m pytest tests/test_proj_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 500 ) > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cp news_recommendation_amd/libnr_engine.so /tmp/libnr_engine_new.so
line() { timeout 600 python bench.py --model NRMS --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'qkv', d['kernel_breakdown_us_per_step'].get('nr_qkv_proj_fwd[S=20]'))"; }
for r in 1 2; do
  cp /tmp/libnr_engine_new.so news_recommendation_amd/libnr_engine.so; line NEW | tee -a $O/ab_qkv_scratch.txt
  cp tools/ab/libnr_engine_old.so news_recommendation_amd/libnr_engine.so; line OLD | tee -a $O/ab_qkv_scratch.txt
done
cp /tmp/libnr_engine_new.so news_recommendation_amd/libnr_engine.so
# the dispatch gap in front of qkv_proj in a graph-replayed step, new build
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --model NRMS --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-train-parity > $O/under_rocprof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_gaps.py $DB 2>&1 | tail -n +2 | head -24 | cut -c1-190 | tee $O/gaps_NRMS_new.txt
rm -rf $O/prof

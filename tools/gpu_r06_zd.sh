#!/bin/bash
# thirtieth GPU pass of round 6: nr_dx_gemm stream form with chunks in straight column order (second version) against the ring, one box
export TMPDIR=/tmp
O=gpurun_out/r06zd
mkdir -p $O
( timeout 300 python -m pytest tests/test_proj_gpu.py -m gpu -x -q -k dx_gemm --timeout 280 ) > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],4), 'value', round(d['value']), {k: v for k, v in kb.items() if 'dx_gemm' in k})"; }
B="--steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for F in 1 0 1 0; do
  NR_DX_STREAM=$F timeout 600 python bench.py --model NRMS $B 2>/dev/null | grep '^{' | tail -1 | ms "NRMS dx_stream=$F (straight order)" | tee -a $O/ab_dx_stream.txt
done

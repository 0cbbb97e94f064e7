#!/bin/bash
# twenty-ninth GPU pass of round 6: nr_dx_gemm as the persistent stream kernel (conv_gemm_kernel<., PLAIN>, NR_DX_STREAM=1) against the
# one-tile-per-workgroup ring (NR_DX_STREAM=0): parity of both forms, NRMS step A/B on one box
export TMPDIR=/tmp
O=gpurun_out/r06zc
mkdir -p $O
( timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 600 ) > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],4), 'value', round(d['value']), {k: v for k, v in kb.items() if 'dx_gemm' in k})"; }
B="--steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for F in 1 0 1 0; do
  NR_DX_STREAM=$F timeout 600 python bench.py --model NRMS $B 2>/dev/null | grep '^{' | tail -1 | ms "NRMS dx_stream=$F" | tee -a $O/ab_dx_stream.txt
done

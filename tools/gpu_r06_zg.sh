#!/bin/bash
# thirty-second GPU pass of round 6: the training forward of the text encoders as gather pass + persistent ring GEMM with the bias / relu / dropout
# epilogue (nr_conv3_fwd_gemm, NR_CONV_FWD_GEMM=1) against the LDS-tile kernel: kernel parity, model parity with the switch on, NAML / LSTUR step A/B
export TMPDIR=/tmp
O=gpurun_out/r06zg
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv_fwd" --timeout 500 ) > $O/pytest_kernels.txt 2>&1; tail -2 $O/pytest_kernels.txt
( NR_CONV_FWD_GEMM=1 timeout 900 python -m pytest tests/test_naml_gpu.py tests/test_lstur_gpu.py -m gpu -x -q --timeout 800 ) > $O/pytest_models.txt 2>&1; tail -2 $O/pytest_models.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],4), 'value', round(d['value']), {k: v for k, v in kb.items() if 'conv3_fwd' in k})"; }
B="--steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for M in NAML LSTUR; do for F in 1 0 1 0; do
  NR_CONV_FWD_GEMM=$F timeout 600 python bench.py --model $M $B 2>/dev/null | grep '^{' | tail -1 | ms "$M fwd_gemm=$F" | tee -a $O/ab_conv_fwd_gemm.txt
done; done

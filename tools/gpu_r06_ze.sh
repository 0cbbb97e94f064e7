#!/bin/bash
# thirty-first GPU pass of round 6: conv data gradient with the chunks of a tile in PAIRS order (both halves of a 128-byte line in consecutive chunks,
# NR_CONVGEMM_PAIRS=1) against tap-inner order, NAML step A/B on one box; parity of the pairs form
export TMPDIR=/tmp
O=gpurun_out/r06ze
mkdir -p $O
( NR_CONVGEMM_PAIRS=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py -m gpu -x -q -k "dgrad or naml or golden" --timeout 500 ) > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_us_per_step']; print('$1 ms', round(d['ms_per_step'],4), 'value', round(d['value']), {k: v for k, v in kb.items() if 'dgrad' in k})"; }
B="--steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-extras"
for F in 1 0 1 0; do
  NR_CONVGEMM_PAIRS=$F timeout 600 python bench.py --model NAML $B 2>/dev/null | grep '^{' | tail -1 | ms "NAML pairs=$F" | tee -a $O/ab_convgemm_pairs.txt
done

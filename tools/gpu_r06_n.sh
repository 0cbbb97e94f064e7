#!/bin/bash
# fourteenth GPU pass of round 6: Wa slot swizzle r & 6 (conflict-free under the bank model) in the pooling backward and forward -- A/B on one box, parity
export TMPDIR=/tmp
O=gpurun_out/r06n
mkdir -p $O
cat > /tmp/ab_cmd.sh <<'EOS'
for K in pool_flat pool_flat_act pool_flat50_act pool_fwd_flat pool_fwd_flat50; do timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done
EOS
bash tools/ab/run_ab.sh bash /tmp/ab_cmd.sh 2>&1 | tee $O/pool_ab.txt
( timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "pool or additive or whole" --timeout 1200 ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
bash tools/pmc_kernel.sh pool_flat50_act pool3_bwd $O/pmc_sq_pool_flat50_act > /dev/null 2>&1
grep "BANK_CONFLICT\|IDX_ACTIVE\|MFMA_BUSY\|GRBM" $O/pmc_sq_pool_flat50_act/summary.txt

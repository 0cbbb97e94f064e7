#!/bin/bash
# thirteenth GPU pass of round 6: pooling backward with the r-form packed tanh arithmetic + ds_add for dq -- A/B against the previous build on one box, parity
export TMPDIR=/tmp
O=gpurun_out/r06m
mkdir -p $O
cat > /tmp/ab_cmd.sh <<'EOS'
for K in pool_flat pool_flat_act pool_flat50_act; do timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done
EOS
bash tools/ab/run_ab.sh bash /tmp/ab_cmd.sh 2>&1 | tee $O/pool3_ab.txt
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_naml_gpu.py tests/test_model_gpu.py -m gpu -q -k "pool or additive or naml or grad or backward" --timeout 1200 ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt

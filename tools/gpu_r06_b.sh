#!/bin/bash
# second GPU pass of round 6: the general-geometry tests, the parity tests again, the training-parity diagnostics
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
( time timeout 1800 python -m pytest tests/test_generic_gpu.py tests/test_trajectory_gpu.py tests/test_training_parity_gpu.py tests/test_config_knobs_gpu.py \
    tests/test_model_gpu.py -m gpu -q --timeout 1200 ) > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt
cp gpurun_out/trajectory_*.json gpurun_out/train_parity_fixture_*.json $O/ 2>/dev/null
( time timeout 1200 python tools/train_parity_diag.py ) > $O/diag.txt 2>&1
grep -v "^$" $O/diag.txt | tail -8

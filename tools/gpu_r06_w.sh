#!/bin/bash
# twenty-third GPU pass of round 6: tests of the fused step against the criterion step
export TMPDIR=/tmp
O=gpurun_out/r06w
mkdir -p $O
( timeout 900 python -m pytest tests/test_step_fused_gpu.py -m gpu -q --timeout 600 ) > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt

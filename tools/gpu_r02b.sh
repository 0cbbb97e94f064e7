#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02b}
mkdir -p $O
timeout 600 python tools/diag_dropin.py > $O/diag_dropin.log 2>&1; cat $O/diag_dropin.log | grep -v amdgpu.ids | head -90
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log

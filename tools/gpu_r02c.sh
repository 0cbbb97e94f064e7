#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02c}
mkdir -p $O
timeout 600 python tools/diag_dropin2.py > $O/diag_dropin2.log 2>&1; grep -v amdgpu.ids $O/diag_dropin2.log | head -60
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log

#!/bin/bash
# Round 3, pass m: phase switches of dx_gemm_ring_kernel (NR_DXR_DEBUG, compile-time variants: 1 no copies after the first three chunks,
# 2 no MFMAs, 4 no stores) and the ring kernels against the two-buffer kernels.
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q -m gpu -k "dx_gemm or tn_gemm" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for d in 0 8 0 8; do
  NR_DXR_DEBUG=$d KB_ONLY=dX_gemm_hand timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | grep dX_gemm_hand | sed "s/^/NR_DXR_DEBUG=$d /" | tee -a $O/dx_ring_phases.txt
done
NR_DX_RING=0 KB_ONLY=dX_gemm_hand timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | grep dX_gemm_hand | sed "s/^/two-buffer kernel /" | tee -a $O/dx_ring_phases.txt
for v in "NR_TN_RING=1" "NR_TN_RING=0"; do
  env $v KB_ONLY=tn_hand timeout 300 python tools/kbench_proj.py 2>/dev/null | grep -E "us$" | sed "s/^/$v /" | tee -a $O/dx_ring_phases.txt
done

#!/bin/bash
# Build libnr_engine.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libnr_engine.so
# -amdgpu-mfma-vgpr-form: MFMA results land in ordinary VGPRs (no v_accvgpr_read copy per accumulator register; the kernels are VALU-bound)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 \
  ${NR_EXTRA_FLAGS:-} nr_engine.hip -o $OUT
echo "built $(realpath $OUT)"

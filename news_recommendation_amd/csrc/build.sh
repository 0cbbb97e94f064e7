#!/bin/bash
# Build libnr_engine.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libnr_engine.so
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-value -Wno-pass-failed ${NR_EXTRA_FLAGS:-}"
mkdir -p _obj
rm -f _obj/nr_engine.o _obj/nr_mhsa2.o          # never link a stale object if a compile fails
# -amdgpu-mfma-vgpr-form: MFMA results land in ordinary VGPRs (no v_accvgpr_read copy per accumulator register; the kernels are
# VALU-bound).  Not for nr_mhsa2.hip: that kernel holds > 256 registers per lane and uses the AGPR half as storage.
$HIPCC $COMMON -mllvm -amdgpu-mfma-vgpr-form=1 -c nr_engine.hip -o _obj/nr_engine.o &
p1=$!
$HIPCC $COMMON -c nr_mhsa2.hip -o _obj/nr_mhsa2.o &
p2=$!
wait $p1      # `wait` without a PID always returns 0: wait for each job so that set -e sees a failed compile
wait $p2
$HIPCC --offload-arch=gfx950 -shared -fPIC _obj/nr_engine.o _obj/nr_mhsa2.o -o $OUT
echo "built $(realpath $OUT)"

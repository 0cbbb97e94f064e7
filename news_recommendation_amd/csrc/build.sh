#!/bin/bash
# Build libnr_engine.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libnr_engine.so
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. -Wno-unused-value \
  ${NR_EXTRA_FLAGS:-} nr_engine.hip -o $OUT
echo "built $(realpath $OUT)"

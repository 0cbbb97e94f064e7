#!/bin/bash
# Build the CPU wave-emulation of the engine's kernels (TEST INFRASTRUCTURE; never loaded by the product).
set -euo pipefail
cd "$(dirname "$0")"
CXX=${NR_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p _build
$CXX -O2 -g -std=c++17 -fPIC -shared -I. -I../../news_recommendation_amd/csrc \
  -Wno-unused-value -Wno-unknown-attributes -Wno-ignored-attributes \
  -x c++ ../../news_recommendation_amd/csrc/nr_engine.hip ../../news_recommendation_amd/csrc/nr_mhsa2.hip nr_emu.cpp -o _build/libnr_engine_emu.so
echo "built $(realpath _build/libnr_engine_emu.so)"

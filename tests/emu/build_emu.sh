#!/bin/bash
# Builds the CPU emulation of libmdm_hip (test infrastructure; see hip_emu.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${MDM_EMU_SO:-$ROOT/build/libmdm_emu.so}   # MDM_EMU_SO / MDM_EMU_CXXFLAGS: variant builds (e.g. -DMDM_X3_RING3)
mkdir -p "$ROOT/build"
"$CXX" -x c++ -std=c++17 -O2 -fPIC -shared -DMDM_EMU -DMDM_PROBES ${MDM_EMU_CXXFLAGS:-} -Wno-psabi -Wno-pass-failed -I"$HERE" -I"$ROOT/motion-diffusion-model_amd/csrc" \
  "$ROOT/motion-diffusion-model_amd/csrc/mdm_api.hip" -o "$OUT"
echo "$OUT"

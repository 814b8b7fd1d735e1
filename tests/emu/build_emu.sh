#!/bin/bash
# Builds the CPU emulation of libmdm_hip (test infrastructure; see hip_emu.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$ROOT/build"
"$CXX" -x c++ -std=c++17 -O2 -fPIC -shared -DMDM_EMU -Wno-psabi -Wno-pass-failed -I"$HERE" -I"$ROOT/motion-diffusion-model_amd/csrc" \
  "$ROOT/motion-diffusion-model_amd/csrc/mdm_api.hip" -o "$ROOT/build/libmdm_emu.so"
echo "$ROOT/build/libmdm_emu.so"

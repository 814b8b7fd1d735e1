#!/bin/bash
# Builds the x3s8 experiment library: `build.sh emu` -> build/libmdm_emu_x3s8.so (CPU emulator, tests: MDM_EMU_SO=...),
# `build.sh gpu` -> build/libmdm_hip_x3s8.so (gfx950; A/B on one box with MDM_HIP_LIB=...).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$ROOT/build"
if [ "${1:-emu}" = emu ]; then
  /opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fPIC -shared -DMDM_EMU -DMDM_PROBES -Wno-psabi -Wno-pass-failed \
    -I"$ROOT/tests/emu" -I"$ROOT/motion-diffusion-model_amd/csrc" -I"$HERE" "$HERE/mdm_api_x3s8.hip" -o "$ROOT/build/libmdm_emu_x3s8.so"
  echo "$ROOT/build/libmdm_emu_x3s8.so"
else
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -fno-slp-vectorize -DMDM_NO_SLP=1 \
    -Rpass-analysis=kernel-resource-usage -I"$ROOT/motion-diffusion-model_amd/csrc" -I"$HERE" "$HERE/mdm_api_x3s8.hip" \
    -o "$ROOT/build/libmdm_hip_x3s8.so" 2> "$ROOT/build/x3s8_resources.txt"
  grep -A12 "gemm_x3s8_kernel" "$ROOT/build/x3s8_resources.txt" | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | head -40
  echo "$ROOT/build/libmdm_hip_x3s8.so"
fi

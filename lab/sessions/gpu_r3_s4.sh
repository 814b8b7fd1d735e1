#!/bin/bash
# Round-3 GPU session 4: bisection builds of the pipelined k-loop against the run-to-run difference of in_proj.
set -u
OUT=gpurun_out/r3s4
mkdir -p $OUT
export TMPDIR=/tmp
for L in build/libmdm_hip_NOLOOK.so build/libmdm_hip_SYNCTILE.so build/libmdm_hip_DRAINBAR.so "$@"; do
  export MDM_HIP_LIB=$PWD/$L
  timeout 300 python tools/gpu_determinism.py 128 10 8 2>&1 | grep -v amdgpu.ids | tail -12 | sed "s|^|[$(basename $L .so)] |"
done | tee $OUT/determinism.txt

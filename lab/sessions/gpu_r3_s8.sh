#!/bin/bash
set -u
OUT=gpurun_out/r3s8
mkdir -p $OUT
export TMPDIR=/tmp
for L in default build/libmdm_hip_probe_SNOP2.so; do
  if [ "$L" = default ]; then unset MDM_HIP_PROBE_LIB; else export MDM_HIP_PROBE_LIB=$PWD/$L; fi
  MDM_X3_PIPE=1 timeout 400 python tools/in_proj_determinism.py 500 256 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300 | sed "s|^|[$(basename $L .so)] |"
done | tee $OUT/in_proj_determinism.txt
unset MDM_HIP_PROBE_LIB
timeout 300 python tools/gpu_determinism.py 128 12 8 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300 | tee $OUT/model_determinism.txt

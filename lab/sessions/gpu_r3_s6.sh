#!/bin/bash
set -u
OUT=gpurun_out/r3s6
mkdir -p $OUT
export TMPDIR=/tmp
for L in default build/libmdm_hip_probe_WDRAIN.so build/libmdm_hip_probe_SNOP.so; do
  if [ "$L" = default ]; then unset MDM_HIP_PROBE_LIB; else export MDM_HIP_PROBE_LIB=$PWD/$L; fi
  MDM_X3_PIPE=1 timeout 400 python tools/in_proj_determinism.py 400 256 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-330 | sed "s|^|[$(basename $L .so)] |"
done | tee $OUT/in_proj_determinism.txt

#!/bin/bash
set -u
OUT=gpurun_out/r3s5
mkdir -p $OUT
export TMPDIR=/tmp
for p in 1 0; do MDM_X3_PIPE=$p timeout 300 python tools/in_proj_determinism.py 30 256 2>&1 | grep -v amdgpu.ids | tail -34; done | tee $OUT/in_proj_determinism.txt
MDM_X3_PIPE=1 timeout 300 python tools/in_proj_determinism.py 30 40 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/in_proj_determinism.txt

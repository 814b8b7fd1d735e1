#!/bin/bash
set -u
OUT=gpurun_out/r3s7
mkdir -p $OUT
export TMPDIR=/tmp
MDM_X3_PIPE=1 timeout 400 python tools/in_proj_determinism.py 300 256 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-700 | sed "s|^|[bias] |" | tee $OUT/in_proj_determinism.txt
PROBE_ZERO_BIAS=1 MDM_X3_PIPE=1 timeout 400 python tools/in_proj_determinism.py 300 256 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-400 | sed "s|^|[zero bias] |" | tee -a $OUT/in_proj_determinism.txt

#!/bin/bash
# Round 3, session 14+: experiments on the DiP concurrent-groups discrepancy (probe library, tools/repro_dip_groups.py, 4 groups
# f16x3, 80 window loops each).  Usage: bash tools/gpu_r3_s14.sh <tag> "ENV1=v ENV2=v" "ENV=v" ...   (one run per argument)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MDM_HIP_LIB=${PROBE_LIB:-$PWD/motion-diffusion-model_amd/csrc/libmdm_hip_probe.so}
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python tools/repro_dip_groups.py ${GROUPS_ARG:-4} ${PREC_ARG:-f16x3} > $OUT/run$i.log 2>&1
  echo "[$cfg]"; grep "FAILS\|ORDER\|DUP" $OUT/run$i.log
done

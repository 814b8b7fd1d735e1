#!/bin/bash
# Round 3, session 13 (time-boxed): does the DiP concurrent-groups discrepancy of round 2 still reproduce on the
# round-3 build?  tools/repro_dip_groups.py on the probe library: 4 groups f16x3, 2 groups f16x3, 4 groups f32 (control).
set -u
OUT=gpurun_out/r3s13
mkdir -p $OUT
export TMPDIR=/tmp
export MDM_HIP_LIB=$PWD/motion-diffusion-model_amd/csrc/libmdm_hip_probe.so
for cfg in "4 f16x3" "2 f16x3" "4 f32"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 400 python tools/repro_dip_groups.py $cfg > $OUT/repro_$tag.log 2>&1
  echo "groups/prec $cfg: $(tail -1 $OUT/repro_$tag.log)"
done

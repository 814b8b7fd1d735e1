#!/bin/bash
# Round 4, session 5 (the ONE time-boxed session on the packed-math co-residency corruption): the DiP window
# loop of SLP-vectorised builds beside torch SDPA on a foreign stream -- plain, and with an explicit wait + idle issue slots between
# the epilogue's LDS patch read and its first (packed) consumer.  The product build (-fno-slp-vectorize) as the control.
set -u
OUT=gpurun_out/${1:-r4s5}
mkdir -p $OUT
for V in "product:" "slp:MDM_HIP_LIB=build/variants/libmdm_slp.so" "slp_nop1:MDM_HIP_LIB=build/variants/libmdm_slp_nop1.so" "slp_nop4:MDM_HIP_LIB=build/variants/libmdm_slp_nop4.so"; do
  TAG=${V%%:*}; ENVS=$(echo "${V#*:}" | tr ',' ' ')
  for R in 1 2; do
    echo "== $TAG round $R"
    env MDM_ALLOW_SLP_BUILD=1 $ENVS timeout 300 python tools/repro_foreign_stream.py f16x3 60 sdpa 2>&1 | grep "differing"
  done
done | tee $OUT/repro.txt

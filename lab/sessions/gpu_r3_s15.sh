#!/bin/bash
# Round 3, session 15: what the 4-wave form of the small split-precision GEMM (-DMDM_X3S_KSPLIT1: immune to the co-residency
# discrepancy of profiles/r03g_dip_groups.md) costs on the DiP bench; same box, interleaved.
set -u
OUT=gpurun_out/r3s35
mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2; do
  for L in default build/libmdm_hip_KS1.so; do
    if [ "$L" = "default" ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$PWD/$L; fi
    N=$(basename $L .so)
    timeout 300 python bench_dip.py --no-cpu-baseline --steps 4 > $OUT/dip_${N}_$r.json 2> $OUT/dip_${N}_$r.err
    python - $OUT/dip_${N}_$r.json $N $r <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:20s} round {sys.argv[3]}: {d['value']:8.2f} {d['unit']}  ms_per_step {d['ms_per_step']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
unset MDM_HIP_LIB

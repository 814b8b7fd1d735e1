#!/bin/bash
# Round 5: cross-CU de-phasing of the production 8-wave GEMM -- probe-library variant (lab/patches/x3_dephase.patch)
set -u
TAG=${1:-r5dephase}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MDM_HIP_LIB=$PWD/build/libmdm_hip_probe_dephase.so
for rep in 1 2; do
  for d in 0 40 80 160 320; do
    MDM_X3_DEPHASE=$d python bench.py --quick --steps 4 --warmup 2 > $OUT/head_d${d}_$rep.json 2> $OUT/head_d${d}_$rep.err
  done
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/head_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]["linear"], d["kernel_ms"]["attention"])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-400:])
PY

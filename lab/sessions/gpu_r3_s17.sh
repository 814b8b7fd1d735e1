#!/bin/bash
# Round 3, session 17: what building WITHOUT SLP vectorisation (no packed fp32 VALU math from the vectorizer: the build that is
# immune to the co-residency discrepancy, profiles/r03g_dip_groups.md) costs -- headline bench and DiP bench, same box, interleaved;
# and the product-flavoured no-SLP library beside the SDPA stream.
set -u
OUT=gpurun_out/r3s45
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s45/ab 2 default build/libmdm_hip_NOSLP.so 2>&1 | tee $OUT/ab.txt
for r in 1 2; do
  for L in default build/libmdm_hip_NOSLP.so; do
    if [ "$L" = "default" ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$PWD/$L; fi
    echo -n "dip $(basename $L .so) round $r: "; timeout 300 python bench_dip.py --no-cpu-baseline --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
  done
done | tee $OUT/dip.txt
unset MDM_HIP_LIB
echo -n "no-SLP product library beside the SDPA stream: "; MDM_HIP_LIB=$PWD/build/libmdm_hip_NOSLP.so timeout 300 python tools/repro_foreign_stream.py f16x3 60 2>&1 | grep sdpa | tee $OUT/sdpa.txt

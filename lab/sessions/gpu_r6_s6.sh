#!/bin/bash
# Round 6, session 6: same-box A/B of the fused hoisted projections, 6 interleaved pairs of 10 generations each (s5 was too noisy).
set -u
TAG=${1:-r6s6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > /dev/null 2>&1    # box warm-up
for i in 1 2 3 4 5 6; do
  for v in before after; do
    if [ $v = after ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_before.so; fi
    python bench_dip.py --steps 10 --warmup 2 --no-cpu-baseline --no-small-batch > $OUT/dip_${v}_$i.json 2> $OUT/dip_${v}_$i.err
  done
done
unset MDM_HIP_LIB
python - $OUT <<'PY'
import json, sys, glob
out = sys.argv[1]
r = {"before": [], "after": []}
for v in r:
    for f in sorted(glob.glob(out + f"/dip_{v}_*.json")):
        r[v].append(json.load(open(f))["value"])
print(r)
import statistics as st
print({v: (round(st.mean(x), 1), round(st.median(x), 1)) for v, x in r.items()}, "ratio of medians", round(st.median(r["after"]) / st.median(r["before"]), 4))
PY

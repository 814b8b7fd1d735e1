#!/bin/bash
# Round 6, session 5: the hoisted memory projections of a DiP window loop as ONE launch for all layers (kv_text, kv_time: 16 -> 2
# launches per window call).  Parity (DiP tests of every round) on the new build, then same-box A/B against the previous build
# (build/variants/libmdm_hip_before.so = HEAD's csrc): B = 32 x3 interleaved, B = 1 / 6 per-call latency.
set -u
TAG=${1:-r6s5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q -x -k "dip or trans_dec or dynamic or DiP" > $OUT/pytest_dip.log 2>&1; tail -2 $OUT/pytest_dip.log
for i in 1 2 3; do
  for v in before after; do
    if [ $v = after ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_before.so; fi
    python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline $([ $i = 1 ] || echo --no-small-batch) > $OUT/dip_${v}_$i.json 2> $OUT/dip_${v}_$i.err
  done
done
unset MDM_HIP_LIB
python - $OUT <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + "/dip_*.json")):
    d = json.load(open(f)); sb = d.get("small_batch")
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], {k: v["window_call_ms"] for k, v in sb.items() if k.startswith("B")} if sb else "")
PY

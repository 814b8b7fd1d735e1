#!/bin/bash
# Small-batch latency sweep: bash tools/gpu_r4_lat.sh <tag> "<B list>" "<variant list>"   (variant = name:ENV=V,ENV=V)
set -u
OUT=gpurun_out/${1:-r4lat}
mkdir -p $OUT
export TMPDIR=/tmp
for B in ${2:-1 6 10}; do for V in ${3:-"auto:"}; do
  TAG=${V%%:*}; ENVS=$(echo "${V#*:}" | tr ',' ' ')
  env $ENVS timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --quick > $OUT/lat_${TAG}_B$B.json 2> $OUT/lat_${TAG}_B$B.err
  python - $OUT/lat_${TAG}_B$B.json $TAG $B <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("latency", sys.argv[2], "B=" + sys.argv[3], d["ms_per_step"], "ms/loop", d["kernel_ms"], "gemm us", d["roofline"]["avg_launch_us"])
except Exception as e:
    print("latency", sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done

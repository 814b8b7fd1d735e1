#!/bin/bash
# Round 4: the four-wave x 64-column, one-wave-per-SIMD form of the pipelined GEMM (MDM_X3_WIDE=1) against the 8-wave form:
# parity subset on the wide form, then interleaved whole-bench A/B.
set -u
OUT=gpurun_out/${1:-r4wide}
mkdir -p $OUT
export TMPDIR=/tmp
MDM_X3_WIDE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "big and f16x3 and (golden or shapes)" > $OUT/pytest_wide.log 2>&1
echo "pytest wide: $(tail -1 $OUT/pytest_wide.log)"
for R in 1 2; do for V in "w8:" "wide:MDM_X3_WIDE=1"; do
  TAG=${V%%:*}; ENVS=$(echo "${V#*:}" | tr ',' ' ')
  env $ENVS timeout 300 python bench.py --steps 3 --warmup 1 --quick > $OUT/ab_${TAG}_$R.json 2> $OUT/ab_${TAG}_$R.err
  python - $OUT/ab_${TAG}_$R.json $TAG $R <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("ab", sys.argv[2], "round", sys.argv[3], d["value"], "motions/s", d["kernel_ms"], "gemm us", d["roofline"]["avg_launch_us"])
except Exception as e:
    print("ab", sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done

#!/bin/bash
# Round 2, GPU visit 2: hostile / envelope / seam tests (no -x: every number is wanted), the DiP tests on the x3 arithmetic,
# the bench line, and a same-box A/B of arithmetic variants and of the sequence-group execution of the encoder.
set -u
TAG=${1:-s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -s -k "hostile or fp16 or const or non_prefix or progressive or config1 or respaced" > $OUT/pytest_r2.log 2>&1
echo "pytest r2 exit $?" >> $OUT/pytest_r2.log
grep -E "^\[parity\]|passed|failed|Error" $OUT/pytest_r2.log | tail -30
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dip or forward_matches_reference_golden or loop_T196" > $OUT/pytest_dip.log 2>&1
echo "pytest dip exit $?" >> $OUT/pytest_dip.log
tail -6 $OUT/pytest_dip.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["unit"], "kernel_ms", d["kernel_ms"]); print("f32_mode", d.get("f32_mode")); print("dip", {k: d["dip"][k] for k in ("value", "ms_per_step", "dtype", "kernel_ms", "roofline")})
PY
tail -3 $OUT/bench.err
BENCH_ARGS="--no-extras" bash tools/gpu_ab.sh $TAG/ab 2 default build/ab/libmdm_bf16.so
ls $OUT

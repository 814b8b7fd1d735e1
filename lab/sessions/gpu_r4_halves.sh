#!/bin/bash
# Round 4: same-box A/B of MDM_ENC_HALVES (in_proj -> attention and linear1 -> linear2 one guidance branch at a time, so that the
# hand-over fits the 256 MB Infinity Cache) on the headline bench; kernel traces of the baseline and of the best candidate.
set -u
TAG=${1:-r4halves}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
run() {
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/h_$tag.json 2> $OUT/h_$tag.err
  python - $OUT/h_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run h0 MDM_ENC_HALVES=0
run h1 MDM_ENC_HALVES=1
run h2 MDM_ENC_HALVES=2
run h3 MDM_ENC_HALVES=3
run h0b MDM_ENC_HALVES=0
run h1b MDM_ENC_HALVES=1
for H in 0 1 3; do
(cd /tmp && MDM_ENC_HALVES=$H timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof$H -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/$OUT/prof$H.json 2> $R/$OUT/prof$H.err)
DB=$(find $OUT/prof$H -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats_h$H.md; cut -c1-190 $OUT/kernel_stats_h$H.md | head -9; rm -f $DB; fi
find $OUT/prof$H -name '*.csv' -size +2M -delete
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden and big" 2>&1 | tail -2
MDM_ENC_HALVES=3 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "full_size or config1 or (golden and big)" 2>&1 | tail -2

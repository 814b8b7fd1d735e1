#!/bin/bash
# One GPU-box visit of round 2: new parity tests first, then the whole GPU suite, the bench line, a same-box A/B of library
# builds and a rocprofv3 kernel trace.  Usage (repo root, on the GPU box): bash tools/gpu_session.sh <tag> [ab-lib ...]
set -u
TAG=${1:-s}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round2.py -m gpu -q -s -x > $OUT/pytest_r2.log 2>&1
echo "pytest r2 exit $?" >> $OUT/pytest_r2.log
grep -E "parity|passed|failed|Error|error" $OUT/pytest_r2.log | tail -40
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round2.py --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 4500 $OUT/bench.json; tail -5 $OUT/bench.err
if [ $# -gt 0 ]; then
  BENCH_ARGS="--no-extras" bash tools/gpu_ab.sh $TAG/ab 2 default "$@"
fi
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md; cut -c1-220 $OUT/kernel_stats.md | head -24; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
ls $OUT

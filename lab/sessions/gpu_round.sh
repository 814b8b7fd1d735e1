#!/bin/bash
# One GPU-box visit: parity tests, the bench line, and a rocprofv3 kernel trace of the same bench command.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [skip-tests]
set -u
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md; cat $OUT/kernel_stats.md; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
ls -la $OUT $OUT/prof 2>/dev/null | head -30

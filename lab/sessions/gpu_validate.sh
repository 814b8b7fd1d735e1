#!/bin/bash
# Full validation visit: every GPU test, smoke(), the default bench line (with extras and the CPU baseline), the one-rank RCCL
# bring-up; optionally ("pmc") the kernel trace + PMC passes.  Usage: bash tools/gpu_validate.sh <tag> [pmc]
set -u
TAG=${1:-v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s --durations=6 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^\.*\[parity\]" $OUT/pytest_gpu.log | sed 's/^\.*//' > $OUT/parity_lines.txt
tail -12 $OUT/pytest_gpu.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1200 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --force-pg --no-extras --no-cpu-baseline --steps 1 > $OUT/bench_pg.json 2> $OUT/bench_pg.err
python -c "
import json; d = json.load(open('$OUT/bench_pg.json')); print('one-rank process group:', d['ranks'], d['value'])" || tail -5 $OUT/bench_pg.err
if [ "${2:-}" = "pmc" ]; then bash tools/gpu_prof.sh $TAG pmc > $OUT/prof.log 2>&1; tail -5 $OUT/prof.log; fi
ls $OUT

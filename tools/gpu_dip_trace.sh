#!/bin/bash
# rocprofv3 kernel trace of the DiP bench (bench_dip.py, 2 timed generations).  Usage: bash tools/gpu_dip_trace.sh <tag>
set -u
TAG=${1:-dip}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 300 python bench_dip.py --steps 3 --no-cpu-baseline --no-small-batch > $OUT/dip.json 2> $OUT/dip.err
python -c "
import json; d = json.load(open('$OUT/dip.json')); print('dip', d['value'], d['ms_per_step'], d['kernel_ms'], d['launches_per_motion_batch'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench_dip.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch > $R/$OUT/prof_dip.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_gaps.py $DB > $OUT/kernel_gaps.md; python tools/rocpd_summary.py $DB --by-grid > $OUT/kernel_stats.md; cut -c1-200 $OUT/kernel_stats.md | head -24; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
tail -c 600 $OUT/prof_dip.json

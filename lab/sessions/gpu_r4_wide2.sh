#!/bin/bash
# per-kernel times of the wide form vs the 8-wave form (one box): rocprofv3 kernel trace of one loop each
set -u
OUT=gpurun_out/${1:-r4wide2}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for V in "w8:" "wide:MDM_X3_WIDE=1"; do
  TAG=${V%%:*}; ENVS=$(echo "${V#*:}" | tr ',' ' ')
  (cd /tmp && env $ENVS timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$TAG -o trace -- python $R/bench.py --steps 1 --warmup 1 --quick > $R/$OUT/prof_$TAG.json 2> $R/$OUT/prof_$TAG.err)
  DB=$(find $OUT/prof_$TAG -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats_$TAG.md; rm -f $DB; fi
  find $OUT/prof_$TAG -name '*.csv' -size +2M -delete
  echo "== $TAG"; head -9 $OUT/kernel_stats_$TAG.md | cut -c1-40,100-200
done

#!/bin/bash
# PMC passes of the DiP bench (bench_dip.py, B = 32 per GPU): kernel trace + FETCH_SIZE / WRITE_SIZE / MFMA busy, separate passes.
set -u
TAG=${1:-r5dippmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench_dip.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch > $R/$OUT/prof_dip.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB --by-grid > $OUT/kernel_stats.md; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for attempt in 1 2; do    # (round 6: one FETCH_SIZE pass died with a SIGSEGV inside the profiled process; a pass that leaves no database is repeated once)
    rm -rf $OUT/pmc$i
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- python $R/bench_dip.py --steps 1 --warmup 0 --no-cpu-baseline --no-small-batch > $R/$OUT/pmc$i.json 2> $R/$OUT/pmc$i.err)
    DB=$(find $OUT/pmc$i -name '*.db' | head -1)
    if [ -n "$DB" ]; then break; fi
  done
  DB=$(find $OUT/pmc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/pmc$i -name '*.csv' -size +1M -delete
done
python tools/dip_pmc_to_json.py $OUT $OUT/dip_pmc.json > $OUT/dip_pmc_to_json.log 2>&1; tail -5 $OUT/dip_pmc_to_json.log

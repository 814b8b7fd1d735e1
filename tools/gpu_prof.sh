#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC passes) of one bench loop. Usage: bash tools/gpu_prof.sh <tag> [pmc]
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
if [ "${SKIP_TRACE:-0}" != "1" ]; then   # SKIP_TRACE=1: PMC passes only (tools/gpu_round.sh already took the kernel trace)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md; cat $OUT/kernel_stats.md | cut -c1-200 | head -16; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
fi
if [ "${2:-}" = "pmc" ]; then
  # counters in their own passes (no trace domains besides --kernel-trace): HBM bytes, MFMA busy, LDS conflicts
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --diffusion-steps 2 > $R/$OUT/pmc$i.json 2> $R/$OUT/pmc$i.err)
    DB=$(find $OUT/pmc$i -name '*.db' | head -1)
    if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
    find $OUT/pmc$i -name '*.csv' -size +1M -delete
    head -40 $OUT/pmc$i.txt
  done
fi
ls $OUT

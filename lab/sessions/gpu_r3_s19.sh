#!/bin/bash
# Round 3, session 19: PMC passes (HBM fetch / write sizes, MFMA busy) of the -fno-slp-vectorize product build; the kernel trace is
# the previous session's (tools/gpu_r3_s12.sh), copied next to them by the caller.
set -u
OUT=gpurun_out/r3s47
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --diffusion-steps 2 > $R/$OUT/pmc$i.json 2> $R/$OUT/pmc$i.err)
  DB=$(find $OUT/pmc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/pmc$i -name '*.csv' -size +1M -delete
  echo "pass $i: $(grep -c '^==' $OUT/pmc$i.txt) kernels"
done

#!/bin/bash
# PMC passes of one small-batch loop: bash tools/gpu_r4_pmc_small.sh <tag> <B> ENV=V...
set -u
TAG=${1:-r4pmcs}; B=${2:-32}; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- python $R/bench.py --batch $B --steps 1 --warmup 0 --quick --diffusion-steps 2 > $R/$OUT/pmc$i.json 2> $R/$OUT/pmc$i.err)
  DB=$(find $OUT/pmc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB gemm_x3 > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/pmc$i -name '*.csv' -size +1M -delete
  head -80 $OUT/pmc$i.txt | cut -c1-150
done

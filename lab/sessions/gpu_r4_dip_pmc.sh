#!/bin/bash
# Round 4: the DiP plane path -- batch sweep of bench_dip.py and PMC passes of one DiP pass (counters in their own runs).
set -u
TAG=${1:-r4dippmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for B in 4 8 16 32 64 256; do
  timeout 300 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > $OUT/dip_B$B.json 2> $OUT/dip_B$B.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['achieved'])" $OUT/dip_B$B.json
done
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- python $R/bench_dip.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/pmc$i.json 2> $R/$OUT/pmc$i.err)
  DB=$(find $OUT/pmc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/pmc$i -name '*.csv' -size +1M -delete
  grep -A9 "gemm_x3s_kernel<2, 1, 8, true, 0, 3\|attention_f32_kernel<1>\|attention_x3_kernel<2" $OUT/pmc$i.txt | head -60 | cut -c1-140
done

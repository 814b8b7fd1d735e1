#!/bin/bash
# rocprofv3 kernel trace + one PMC pass (MFMA busy) of tools/gemm_probe.py: the f16f6 k-loop next to the f16x3 kernel at the
# encoder's GEMM shapes.  Usage: bash tools/gpu_prof_f6.sh <tag>
set -u
TAG=${1:-f6prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && PROBE_WAVES=8 timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/tools/gemm_probe.py 5 0 > $R/$OUT/probe.txt 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md; cut -c1-220 $OUT/kernel_stats.md | head -24; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete
(cd /tmp && PROBE_WAVES=8 timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$OUT/pmc1 -o pmc -- python $R/tools/gemm_probe.py 2 0 > $R/$OUT/pmc1.out 2> $R/$OUT/pmc1.err)
DB=$(find $OUT/pmc1 -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc1.txt 2>&1; rm -f $DB; fi
find $OUT/pmc1 -name '*.csv' -size +1M -delete
grep -A4 "gemm_x3_kernel" $OUT/pmc1.txt | cut -c1-200 | head -80

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration by request width (tools/pmc_calib/pmc_calib.hip): separate --pmc passes, kernel trace only.
set -u
TAG=${1:-r5calib}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
[ -x build/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib/pmc_calib.hip -o build/pmc_calib
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*" | sort -u | tr '\n' ' ') > $OUT/tcc_counters.txt 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/pmc$i -o pmc -- $R/build/pmc_calib 1024 > $R/$OUT/calib$i.json 2> $R/$OUT/calib$i.err)
  DB=$(find $OUT/pmc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB > $OUT/pmc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/pmc$i -name '*.csv' -size +1M -delete
  cat $OUT/pmc$i.txt | head -30
done
cat $OUT/calib1.json; head -c 600 $OUT/tcc_counters.txt

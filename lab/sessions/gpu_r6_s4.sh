#!/bin/bash
# Round 6, session 4: (a) the capture-refusal test (now with the mask pre-classified); (b) A/B of the plane stores' cache policy on the
# DiP bench -- default / non-temporal / write-through variant builds (build/variants/, -DMDM_STORE_VARIANT=1|2), interleaved twice,
# + inter-kernel gaps of each; (c) headline marker per variant.
set -u
TAG=${1:-r6s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -s -k "first_use" > $OUT/pytest_capture.log 2>&1; tail -3 $OUT/pytest_capture.log; grep "EXC\|REFUSED\|REPLAY" $OUT/pytest_capture.log | head
for i in 1 2; do
  for v in default nt wt; do
    if [ $v = default ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_$v.so; fi
    python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > $OUT/dip_${v}_$i.json 2> $OUT/dip_${v}_$i.err
  done
done
for v in default nt wt; do
  if [ $v = default ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_$v.so; fi
  python bench.py --quick > $OUT/head_$v.json 2> $OUT/head_$v.err
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_$v -o trace -- python $R/bench_dip.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch > $R/$OUT/prof_$v.json 2> $R/$OUT/prof_$v.err)
  DB=$(find $OUT/prof_$v -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_gaps.py $DB > $OUT/gaps_$v.md; python tools/rocpd_summary.py $DB --by-grid > $OUT/kernel_stats_$v.md; rm -f $DB; fi
  find $OUT/prof_$v -name '*.csv' -size +1M -delete
done
unset MDM_HIP_LIB
python - $OUT <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + "/dip_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]["linear"])
for f in sorted(glob.glob(out + "/head_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
for f in sorted(glob.glob(out + "/gaps_*.md")):
    print(f.split("/")[-1], open(f).readline().strip()[:230])
PY

#!/bin/bash
# Round 4, session 1: the new parity tests on the hardware, small-batch latency baselines (both arithmetic modes), and the
# L2 hit / miss counters of the encoder GEMMs (are in_proj's 2.7x reads weight re-fetches?).
set -u
OUT=gpurun_out/${1:-r4s1}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q > $OUT/pytest_r4.log 2>&1
echo "pytest r4: $(tail -1 $OUT/pytest_r4.log)"
grep "parity\]" $OUT/pytest_r4.log > $OUT/parity_lines.txt; cat $OUT/parity_lines.txt | cut -c1-160
for P in f16x3 f32; do for B in 1 6 10 32; do
  timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --quick --precision $P > $OUT/lat_${P}_B$B.json 2> $OUT/lat_${P}_B$B.err
  python - $OUT/lat_${P}_B$B.json $P $B <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("latency", sys.argv[2], "B=" + sys.argv[3], d["ms_per_step"], "ms/loop", d["kernel_ms"])
except Exception as e:
    print("latency", sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done
# TCC hit / miss / fabric read requests of one 2-step loop (counters in their own pass, kernel-trace only)
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/tcc$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --quick --diffusion-steps 2 > $R/$OUT/tcc$i.json 2> $R/$OUT/tcc$i.err)
  DB=$(find $OUT/tcc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB gemm_x3 > $OUT/tcc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/tcc$i -name '*.csv' -size +1M -delete
  head -60 $OUT/tcc$i.txt | cut -c1-170
done
timeout 300 python bench.py --quick > $OUT/bench_quick.json 2> $OUT/bench_quick.err; cat $OUT/bench_quick.json | cut -c1-600

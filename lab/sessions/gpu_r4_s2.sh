#!/bin/bash
# Round 4, session 2: the small-row-count GEMM (csrc/gemm_x3s.h) on the hardware: parity on both kernels, latency sweep over
# batch sizes and tile heights against gemm_x3.h's sequence tiles, and the 128-byte fabric request counter (TCC_BUBBLE).
set -u
OUT=gpurun_out/${1:-r4s2}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "not config1 and not config2 and not dip and not f16f6" > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"
grep "parity\]" $OUT/pytest.log > $OUT/parity_lines.txt
for B in 1 6 10 16 32 64; do for V in "small1:MDM_X3S_RT=1 MDM_X3S_MAX_SEQS=999" "small2:MDM_X3S_RT=2 MDM_X3S_MAX_SEQS=999" "big:MDM_X3S_MAX_SEQS=0"; do
  TAG=${V%%:*}; ENVS=${V#*:}
  env $ENVS timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --quick > $OUT/lat_${TAG}_B$B.json 2> $OUT/lat_${TAG}_B$B.err
  python - $OUT/lat_${TAG}_B$B.json $TAG $B <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("latency", sys.argv[2], "B=" + sys.argv[3], d["ms_per_step"], "ms/loop", d["kernel_ms"], "gemm us", d["roofline"]["avg_launch_us"])
except Exception as e:
    print("latency", sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done
i=0
for C in "TCC_EA0_RDREQ_sum TCC_BUBBLE_sum" ; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d $R/$OUT/tcc$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --quick --diffusion-steps 2 > $R/$OUT/tcc$i.json 2> $R/$OUT/tcc$i.err)
  DB=$(find $OUT/tcc$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_pmc.py $DB gemm_x3 > $OUT/tcc$i.txt 2>&1; rm -f $DB; fi
  find $OUT/tcc$i -name '*.csv' -size +1M -delete
  head -60 $OUT/tcc$i.txt | cut -c1-170
done

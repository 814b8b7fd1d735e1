#!/bin/bash
# Round 5: the 32- / 64-row tile cross-over of gemm_x3s_kernel re-measured on the final kernels (default: 32 rows up to 12 sequences),
# and the new defaults (row tiles up to 80 sequences; one-kernel cross-attention block from 144 tiles) confirmed at B = 24 / DiP B = 40.
set -u
TAG=${1:-r5xover3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --quick --steps 3 --warmup 1 > $OUT/enc_b128_marker.json 2> $OUT/enc_b128_marker.err   # the box-speed marker
for b in 4 6 8 10 12; do
  for rt in 1 2; do
    python bench.py --quick --batch $b --steps 5 --warmup 2 --engine-option small_gemm_row_tiles=$rt > $OUT/enc_b${b}_rt$rt.json 2> $OUT/enc_b${b}_rt$rt.err
  done
done
python bench.py --quick --batch 24 --steps 4 --warmup 2 > $OUT/enc_b24_default.json 2> $OUT/enc_b24_default.err
python bench.py --quick --batch 32 --steps 4 --warmup 2 > $OUT/enc_b32_default.json 2> $OUT/enc_b32_default.err
python bench_dip.py --batch 40 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/dip_b40_default.json 2> $OUT/dip_b40_default.err
python bench_dip.py --batch 32 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/dip_b32_default.json 2> $OUT/dip_b32_default.err
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]["linear"], d["kernel_ms"].get("attention"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-300:])
PY

#!/bin/bash
# Round 5, after the v_fma_mix operand split (short kernels +7 %): where do the cross-overs sit now?
#   encoder: gemm_x3s_kernel (row tiles) against gemm_x3_kernel (sequence tiles) at B = 16 ... 40 (default: row tiles up to 40 sequences = B 20)
#   DiP: the cross-attention block forms (MDM_OPT_DEC_FUSED_XATTN 1 / 2; default 3 = by size, cross-over at 160 tiles) at B = 32 / 48 / 64
set -u
TAG=${1:-r5xover}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --quick --steps 3 --warmup 1 > $OUT/enc_b128_marker.json 2> $OUT/enc_b128_marker.err   # the box-speed marker
for b in 20 24 32 40 48 64; do
  python bench.py --quick --batch $b --steps 4 --warmup 2 --engine-option small_gemm_max_seqs=0 > $OUT/enc_b${b}_big.json 2> $OUT/enc_b${b}_big.err
  python bench.py --quick --batch $b --steps 4 --warmup 2 --engine-option small_gemm_max_seqs=128 > $OUT/enc_b${b}_small.json 2> $OUT/enc_b${b}_small.err
done
for b in 32 40 48; do
  for x in 1 2; do
    python bench_dip.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --xattn $x > $OUT/dip_b${b}_x$x.json 2> $OUT/dip_b${b}_x$x.err
  done
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]["linear"], d["kernel_ms"].get("attention"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-300:])
PY

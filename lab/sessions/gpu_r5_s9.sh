#!/bin/bash
# Round 5, session 9: seq-head blocks with the first fill mapping restored + unused key tile skipped: DiP A/B + timeline.
set -u
TAG=${1:-r5s9}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -x -k "fused" > $OUT/pytest_fused.log 2>&1
echo "pytest fused: $(tail -1 $OUT/pytest_fused.log)"; grep "FAILED\|Error" $OUT/pytest_fused.log | head
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_auto_$i.json 2> $OUT/dip_auto_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --xattn 1 > $OUT/dip_x1_$i.json 2> $OUT/dip_x1_$i.err
done
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 48 --xattn 1 > $OUT/dip_x1_B48.json 2> $OUT/dip_x1_B48.err
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 48 --xattn 2 > $OUT/dip_x2_B48.json 2> $OUT/dip_x2_B48.err
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 --xattn 1 > $OUT/dip_x1_B64.json 2> $OUT/dip_x1_B64.err
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 --xattn 2 > $OUT/dip_x2_B64.json 2> $OUT/dip_x2_B64.err
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], d["kernel_ms"]["linear"])
PY
timeout 200 python tools/sb_timeline.py 32 > $OUT/sb_timeline.txt 2>&1; tail -16 $OUT/sb_timeline.txt | cut -c1-200

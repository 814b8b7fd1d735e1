#!/bin/bash
# Round 5, session 3: xattn_block_kernel v2 (4-slot W ring, transposed out_proj without patch round trips): parity, same-box A/B,
# per-phase timeline of one launch (probe library), kernel trace.
set -u
TAG=${1:-r5s3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -x -k "fused" > $OUT/pytest_fused.log 2>&1
echo "pytest fused: $(tail -1 $OUT/pytest_fused.log)"; grep -o "\[parity\].*" $OUT/pytest_fused.log | head -20; grep "FAILED\|Error" $OUT/pytest_fused.log | head
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_fused_$i.json 2> $OUT/dip_fused_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused-xattn > $OUT/dip_three_$i.json 2> $OUT/dip_three_$i.err
done
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 > $OUT/dip_fused_B64.json 2> $OUT/dip_fused_B64.err
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], d["kernel_ms"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 200 python tools/xb_timeline.py 32 > $OUT/xb_timeline.txt 2>&1; cat $OUT/xb_timeline.txt | tail -22
bash tools/gpu_dip_trace.sh $TAG/trace > $OUT/trace.log 2>&1; head -9 gpurun_out/$TAG/trace/kernel_stats.md 2>/dev/null | cut -c1-200

#!/bin/bash
# Round 5, session 5: the cross-attention as a (sequence, head) kernel (selfattn_block.h CROSS) + out_proj GEMM, against the one-kernel
# block and the three-launch form: parity, same-box A/B, timelines of the self / cross launches, kernel trace; PMC calibration.
set -u
TAG=${1:-r5s5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -x -k "fused or dip" > $OUT/pytest_fused.log 2>&1
echo "pytest fused: $(tail -1 $OUT/pytest_fused.log)"; grep -o "\[parity\].*" $OUT/pytest_fused.log | grep "sequence, head" | head -12; grep "FAILED\|Error" $OUT/pytest_fused.log | head
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_x2_$i.json 2> $OUT/dip_x2_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --xattn 1 > $OUT/dip_x1_$i.json 2> $OUT/dip_x1_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --xattn 0 > $OUT/dip_x0_$i.json 2> $OUT/dip_x0_$i.err
done
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 > $OUT/dip_x2_B64.json 2> $OUT/dip_x2_B64.err
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 128 > $OUT/dip_x2_B128.json 2> $OUT/dip_x2_B128.err
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], d["kernel_ms"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 200 python tools/sb_timeline.py 32 > $OUT/sb_timeline.txt 2>&1; tail -16 $OUT/sb_timeline.txt | cut -c1-200
bash tools/gpu_dip_trace.sh $TAG/trace > $OUT/trace.log 2>&1; head -12 gpurun_out/$TAG/trace/kernel_stats.md 2>/dev/null | cut -c1-200
bash tools/gpu_r5_pmc_calib.sh $TAG/calib > $OUT/calib.log 2>&1; tail -60 $OUT/calib.log | cut -c1-200

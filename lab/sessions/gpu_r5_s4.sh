#!/bin/bash
# Round 5, session 4: selfattn_block_kernel (in_proj + self-attention per (sequence, head)) on top of xattn_block_kernel v2: parity,
# same-box A/B of the four combinations, kernel trace.
set -u
TAG=${1:-r5s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -x -k "fused or dip" > $OUT/pytest_fused.log 2>&1
echo "pytest fused: $(tail -1 $OUT/pytest_fused.log)"; grep -o "\[parity\].*" $OUT/pytest_fused.log | head -30; grep "FAILED\|Error" $OUT/pytest_fused.log | head
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_both_$i.json 2> $OUT/dip_both_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused-selfattn > $OUT/dip_xonly_$i.json 2> $OUT/dip_xonly_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused-xattn > $OUT/dip_saonly_$i.json 2> $OUT/dip_saonly_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused-xattn --no-fused-selfattn > $OUT/dip_none_$i.json 2> $OUT/dip_none_$i.err
done
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64 > $OUT/dip_both_B64.json 2> $OUT/dip_both_B64.err
python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 128 > $OUT/dip_both_B128.json 2> $OUT/dip_both_B128.err
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], d["kernel_ms"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "unreadable", e)
PY
bash tools/gpu_dip_trace.sh $TAG/trace > $OUT/trace.log 2>&1; head -12 gpurun_out/$TAG/trace/kernel_stats.md 2>/dev/null | cut -c1-200
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_zz_coresidency.py -m gpu -q -s -k "dip or trans_dec" > $OUT/pytest_dip.log 2>&1
echo "pytest dip subset: $(tail -1 $OUT/pytest_dip.log)"; grep "FAILED\|Error" $OUT/pytest_dip.log | head

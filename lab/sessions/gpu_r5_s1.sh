#!/bin/bash
# Round 5, session 1: the masked DiP plane route, handle options, wide latent dims, graph capture -- parity first, then the DiP bench on
# the --mask_frames recipe (same-box A/B against the unmasked model) and a headline sanity line.
set -u
TAG=${1:-r5s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -s > $OUT/pytest_r5.log 2>&1
echo "pytest r5: $(tail -1 $OUT/pytest_r5.log)"; grep "^\[parity\]\|FAILED\|Error" $OUT/pytest_r5.log | head -40
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_masked_$i.json 2> $OUT/dip_masked_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-mask-frames > $OUT/dip_plain_$i.json 2> $OUT/dip_plain_$i.err
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"], d["kernel_ms"])
    except Exception as e:
        print(f, "unreadable", e)
PY
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench_quick.json 2> $OUT/bench_quick.err
python -c "
import json,sys
d=json.load(open('$OUT/bench_quick.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -q -s -k "dip or small_batch or forward_matches" > $OUT/pytest_dip.log 2>&1
echo "pytest dip subset: $(tail -1 $OUT/pytest_dip.log)"; grep "FAILED\|Error" $OUT/pytest_dip.log | head

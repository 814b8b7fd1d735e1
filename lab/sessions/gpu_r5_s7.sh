#!/bin/bash
# Round 5, session 7: attention_x3 DIRECT (planes from the accumulators, next item's tiles in front of the stores) on the headline:
# parity (encoder goldens on both forms), same-box A/B, kernel trace of the winner; seq-head blocks v2b on the DiP bench.
set -u
TAG=${1:-r5s7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -s -x -k "attention_direct" > $OUT/pytest_direct.log 2>&1
echo "pytest direct: $(tail -1 $OUT/pytest_direct.log)"; grep -o "\[parity\].*" $OUT/pytest_direct.log | head; grep "FAILED\|Error" $OUT/pytest_direct.log | head
for i in 1 2; do
  python bench.py --quick --steps 6 --warmup 2 > $OUT/head_staged_$i.json 2> $OUT/head_staged_$i.err
  python bench.py --quick --steps 6 --warmup 2 --engine-option attn_direct_out=1 > $OUT/head_direct_$i.json 2> $OUT/head_direct_$i.err
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/head_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"])
    except Exception as e:
        print(f, "unreadable", e)
PY
for i in 1 2; do
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_auto_$i.json 2> $OUT/dip_auto_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --xattn 1 > $OUT/dip_x1_$i.json 2> $OUT/dip_x1_$i.err
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/dip_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["launches_per_motion_batch"])
PY
timeout 200 python tools/sb_timeline.py 32 > $OUT/sb_timeline.txt 2>&1; tail -16 $OUT/sb_timeline.txt | cut -c1-200
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --quick --engine-option attn_direct_out=1 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/kernel_stats_direct.md; head -10 $OUT/kernel_stats_direct.md | cut -c1-200; rm -f $DB; fi
find $OUT/prof -name '*.csv' -size +2M -delete

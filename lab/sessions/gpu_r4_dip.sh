#!/bin/bash
# Round 4, DiP decoder on operand planes: (1) the DiP / trans_dec GPU tests on every route, (2) same-box A/B of bench_dip.py
# (planes on 64- / 32-row tiles vs the fp32 skeleton, B = 32 and B = 64), (3) kernel trace of one DiP pass, (4) the closing
# evidence of the product build: kernel trace + PMC passes of the bench command and the full bench line.
set -u
TAG=${1:-r4dip}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
timeout 900 python -m pytest tests -m gpu -x -q -s -k "dip or trans_dec or DiP" --durations=8 > $OUT/pytest_dip.log 2>&1
echo "pytest dip: $(tail -1 $OUT/pytest_dip.log)"
grep "parity\]" $OUT/pytest_dip.log | sed 's/^[.s]*//' > $OUT/parity_lines_dip.txt; wc -l $OUT/parity_lines_dip.txt
run() {   # tag, env..., -- args
  local tag=$1; shift
  env "$@" timeout 300 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch ${BATCH:-32} > $OUT/dip_$tag.json 2> $OUT/dip_$tag.err
  python - $OUT/dip_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["launches"], d["roofline"]["avg_launch_us"], d["launches_per_motion_batch"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run planes A=1
run planes_rt1 MDM_X3S_RT=1
run skeleton MDM_X3S_MAX_SEQS=0
BATCH=64 run planes_B64 A=1
BATCH=64 run skeleton_B64 MDM_X3S_MAX_SEQS=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/dipprof -o trace -- python $R/bench_dip.py --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/dipprof.json 2> $R/$OUT/dipprof.err)
DB=$(find $OUT/dipprof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > $OUT/dip_kernel_stats.md; cut -c1-200 $OUT/dip_kernel_stats.md | head -18; rm -f $DB; fi
find $OUT/dipprof -name '*.csv' -size +2M -delete
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/gpu_prof.sh $TAG/prof pmc > $OUT/prof.log 2>&1
head -8 $OUT/prof/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"])
PY

#!/bin/bash
# Round 3, session 18: the -fno-slp-vectorize build (the product from here on): smoke, a parity subset incl. the DiP tests, the bench line.
set -u
OUT=gpurun_out/r3s46
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
K="forward_matches_reference_golden or loop_matches_reference_golden or dip or mdm_linear_x3 or shard_invariance or config1"
timeout 200 python -m pytest tests -m gpu -x -q -s -k "$K" > $OUT/pytest_subset.log 2>&1
echo "pytest subset: $(tail -1 $OUT/pytest_subset.log)"
grep "parity\]" $OUT/pytest_subset.log > $OUT/parity_lines.txt
timeout 200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"])
PY

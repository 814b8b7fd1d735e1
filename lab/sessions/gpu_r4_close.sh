#!/bin/bash
# Round 4, last session: the product build after the RES_LDS experiment was taken out again (= the sources of r4final3 + probe-only
# code): kernel trace + PMC passes of the bench command, the full bench line, smoke, the golden / DiP-route GPU tests.
set -u
TAG=${1:-r4close}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
bash tools/gpu_prof.sh $TAG/prof pmc > $OUT/prof.log 2>&1
head -7 $OUT/prof/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"])
PY
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -x -q -s -k "golden" > $OUT/pytest_subset.log 2>&1
echo "pytest subset: $(tail -1 $OUT/pytest_subset.log)"

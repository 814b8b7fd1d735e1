#!/bin/bash
# Round-3 closing session on the final sources: full GPU suite, kernel trace + PMC passes of the bench command, the bench line.
set -u
OUT=gpurun_out/${1:-r3final}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256())" > $OUT/csrc_sha256.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log > $OUT/parity_lines.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/gpu_prof.sh ${1:-r3final}/prof pmc > $OUT/prof.log 2>&1
head -14 $OUT/prof/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY

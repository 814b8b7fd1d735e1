#!/bin/bash
# Round-3 GPU session 9: the fixed pipelined loop -- full GPU suite, same-box A/B, kernel trace + PMC passes, full bench line.
set -u
OUT=gpurun_out/r3s9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log > $OUT/parity_lines.txt
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s9/ab 2 env:MDM_X3_PIPE=0 env:MDM_X3_PIPE=1
bash tools/gpu_prof.sh r3s9/prof pmc > $OUT/prof.log 2>&1
tail -30 $OUT/prof.log | cut -c1-220
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 600 $OUT/bench_full.json

#!/bin/bash
# Round-3 GPU session 1: parity of the pipelined k-loop (full GPU suite), same-box A/B against the step-synchronous loop,
# kernel trace.  Usage (from the repo root on the GPU box): bash tools/gpu_r3_s1.sh
set -u
OUT=gpurun_out/r3s1
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py build > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest(PIPE=1): $(tail -1 $OUT/pytest_gpu.log)"
MDM_X3_PIPE=0 timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "config1 or config2 or hostile" > $OUT/pytest_gpu_nopipe.log 2>&1
echo "pytest(PIPE=0 subset): $(tail -1 $OUT/pytest_gpu_nopipe.log)"
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s1/ab 2 env:MDM_X3_PIPE=0 env:MDM_X3_PIPE=1
bash tools/gpu_prof.sh r3s1/prof
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 1500 $OUT/bench_full.json

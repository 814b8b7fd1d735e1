#!/bin/bash
# One GPU-box visit for kernel experiments: probes first (cheap), then a parity subset, then the bench line.
# Usage: bash tools/gpu_exp.sh <tag> [pytest -k expression]
set -u
TAG=${1:-exp}
KEXPR=${2:-"attention or linear_f16x3 or forward_matches or loop_matches"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/gemm_probe.py 10 ${ABLS:-0,1,2,4,8,16,17,24} > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json; tail -5 $OUT/bench.err

#!/bin/bash
# Round 3, session 11: clipped-window epilogue (default build) and rational-erf GELU (build/libmdm_hip_GELUR.so) against the
# previous build (build/libmdm_hip_BASE.so): quick parity of both, then same-box A/B.
set -u
OUT=gpurun_out/r3s11
mkdir -p $OUT
export TMPDIR=/tmp
K="mdm_linear_x3 or forward_matches_reference_golden or loop_matches_reference_golden or shard_invariance or config1"
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest_default.log 2>&1
echo "parity default: $(tail -1 $OUT/pytest_default.log)"
PARITY_K="$K" BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s11/ab 2 build/libmdm_hip_BASE.so default build/libmdm_hip_GELUR.so 2>&1 | tee $OUT/ab.txt

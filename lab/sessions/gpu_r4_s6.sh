#!/bin/bash
# contiguous row tiling of the small GEMMs: parity subset + latency
set -u
OUT=gpurun_out/${1:-r4s6}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -k "small" > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"
bash tools/gpu_r4_lat.sh ${1:-r4s6} "1 2 4 6 10 16 20" "auto:"

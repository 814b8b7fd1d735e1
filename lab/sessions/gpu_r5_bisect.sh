#!/bin/bash
# Round 5: which change broke xattn_block_kernel<4, 3> (70 memory tokens)?  The failing GPU tests on variant builds of the same sources.
set -u
TAG=${1:-r5bisect}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
K="2-0-64 or 3-0-64"
for v in product pksub nop one noahead; do
  if [ $v = product ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$PWD/build/v_$v/libmdm_hip.so; fi
  timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -m gpu -q -s -k "$K" > $OUT/pytest_$v.log 2>&1
  echo "$v: $(tail -1 $OUT/pytest_$v.log)"; grep -o "\[parity\].*" $OUT/pytest_$v.log | cut -c1-200
done

#!/bin/bash
# Round-3 GPU session 3: (1) do LDS-DMA and VGPR loads retire in issue order on vmcnt?  (2) pipelined loop with waits that do not
# assume it (STRICT): determinism + throughput  (3) time decomposition of the two k-loops on 208-row tiles.
set -u
OUT=gpurun_out/r3s3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 build/vmcnt_order 2>&1 | tee $OUT/vmcnt_order.txt
for L in default build/libmdm_hip_strict.so; do
  if [ "$L" = default ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$PWD/$L; fi
  timeout 300 python tools/gpu_determinism.py 128 12 8 2>&1 | tail -14 | sed "s|^|[$L] |"
done | tee $OUT/determinism.txt
unset MDM_HIP_LIB
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s3/ab 2 env:MDM_X3_PIPE=0 default build/libmdm_hip_strict.so 2>&1 | tee $OUT/ab.txt
timeout 600 python tools/gemm_probe_pipe.py 10 0,2,4,7 > $OUT/gemm_probe_pipe.txt 2>&1
cat $OUT/gemm_probe_pipe.txt

#!/bin/bash
# Round 3, session 16: foreign-stream contention (torch SDPA kernels on a second stream) -- the encoder path, and the DiP path with the
# 4-wave build of the small GEMM.
set -u
OUT=gpurun_out/r3s37
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/repro_foreign_encoder.py 32 40 > $OUT/encoder.log 2>&1; grep "encoder forward" $OUT/encoder.log
MDM_HIP_LIB=$PWD/build/libmdm_hip_KS1.so timeout 300 python tools/repro_foreign_stream.py f16x3 40 > $OUT/dip_ks1.log 2>&1; grep "one chain" $OUT/dip_ks1.log

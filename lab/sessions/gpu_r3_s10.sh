#!/bin/bash
set -u
OUT=gpurun_out/r3s10
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s10/ab 2 env:MDM_X3_PIPE=0 default build/libmdm_hip_NOGUARD.so build/libmdm_hip_DEPTH3.so build/libmdm_hip_PRIO.so 2>&1 | tee $OUT/ab.txt

#!/bin/bash
# Round-3 GPU session 2: localise the run-to-run difference of the pipelined k-loop, decompose its time.
set -u
OUT=gpurun_out/r3s2
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py build > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
for cfg in "0 63" "1 63" "1 1" "1 4" "1 8" "1 16" "1 32" "1 2"; do
  set -- $cfg
  MDM_X3_PIPE=$1 MDM_X3_PIPE_KINDS=$2 timeout 300 python tools/gpu_determinism.py 128 6 8 2>&1 | tail -8 | sed "s/^/[pipe=$1 kinds=$2] /"
done | tee $OUT/determinism.txt
MDM_X3_PIPE=1 timeout 300 python tools/gpu_determinism.py 16 6 8 2>&1 | tail -3 | sed "s/^/[pipe=1 B=16] /" | tee -a $OUT/determinism.txt
MDM_X3_PIPE=1 timeout 300 python tools/gpu_determinism.py 128 6 1 2>&1 | tail -3 | sed "s/^/[pipe=1 layers=1] /" | tee -a $OUT/determinism.txt
timeout 600 python tools/gemm_probe_pipe.py 10 > $OUT/gemm_probe_pipe.txt 2>&1
cat $OUT/gemm_probe_pipe.txt

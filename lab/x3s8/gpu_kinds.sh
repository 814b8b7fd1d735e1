#!/bin/bash
# x3s8 experiment: which GEMM kinds lose on the 8-wave form?  bench_dip.py with MDM_X3S8_KINDS = 16 (cross q-projection / OutputProcess),
# 4 (out_proj, cross out_proj, linear2), 0 (none: gemm_x3s_kernel through the experiment library), 22 (default)
set -u
OUT=gpurun_out/${1:-r4x3s8k}; mkdir -p $OUT
X=$PWD/build/libmdm_hip_x3s8.so
for k in 16 4 0 22; do
  MDM_HIP_LIB=$X MDM_X3S8_KINDS=$k timeout 12 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 32 > $OUT/dip_k$k.json 2>> $OUT/err.txt
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print('kinds', sys.argv[2], d['value'], d['ms_per_step'], d['kernel_ms']['linear'])" $OUT/dip_k$k.json $k | tee -a $OUT/kinds.txt
done

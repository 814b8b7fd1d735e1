#!/bin/bash
# x3s8 experiment: parity of the experiment library on the GPU, then same-box A/B of bench_dip.py (experiment / product, interleaved)
set -u
OUT=gpurun_out/${1:-r4x3s8}; mkdir -p $OUT
X=$PWD/build/libmdm_hip_x3s8.so
MDM_HIP_LIB=$X MDM_X3S_RT=2 timeout 20 python tools/x3s8/check_gpu.py 2>&1 | tail -1 | tee $OUT/parity.txt
for v in x3s8 product x3s8 product; do
  if [ $v = x3s8 ]; then E="MDM_HIP_LIB=$X"; else E="A=1"; fi
  env $E timeout 15 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 32 > $OUT/dip_$v.json 2>> $OUT/err.txt
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['kernel_ms']['linear'], d['roofline']['avg_launch_us'])" $OUT/dip_$v.json $v | tee -a $OUT/ab.txt
done

#!/bin/bash
# Round 5: the operand split without packed fp32 math (v_fma_mix, the new default) against the two-step form (v_pk_add_f32,
# build/libmdm_hip_pksub.so = the same sources with -DMDM_SPLIT_PKSUB): bit-exactness tests, same-box A/B on both benches.
set -u
TAG=${1:-r5mix}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_zz_coresidency.py -m gpu -q -s -k "operand_split or loop_matches_reference_golden or dip_forward_matches_reference or coresidency or foreign" > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep "FAILED\|Error" $OUT/pytest.log | head
for i in 1 2; do
  python bench.py --quick --steps 6 --warmup 2 > $OUT/head_mix_$i.json 2> $OUT/head_mix_$i.err
  MDM_HIP_LIB=$PWD/build/libmdm_hip_pksub.so python bench.py --quick --steps 6 --warmup 2 > $OUT/head_pksub_$i.json 2> $OUT/head_pksub_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_mix_$i.json 2> $OUT/dip_mix_$i.err
  MDM_HIP_LIB=$PWD/build/libmdm_hip_pksub.so python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_pksub_$i.json 2> $OUT/dip_pksub_$i.err
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"])
    except Exception as e:
        print(f, "unreadable", e)
PY

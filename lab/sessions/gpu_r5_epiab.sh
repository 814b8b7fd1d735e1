#!/bin/bash
# Round 5: the epilogues' patch read one more round ahead (gemm_x3.h MDM_X3_EPI_AHEAD, gemm_x3s.h MDM_X3S_EPI_AHEAD; variant
# build/libmdm_hip_epis.so = the same sources with both switches on) against the product library: parity tests on the variant,
# then same-box A/B on the headline, the DiP bench and the latency regime, interleaved, two passes.
set -u
TAG=${1:-r5epi}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/build/libmdm_hip_epis.so
MDM_HIP_LIB=$V timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -q -s -x \
  -k "forward_matches_reference_golden or loop_matches_reference_golden or mdm_linear_x3 or dip_forward_matches_reference or small_batch_loop or clip_denoised_and_fixed or wide_latent" > $OUT/pytest.log 2>&1
echo "pytest (variant): $(tail -1 $OUT/pytest.log)"; grep "FAILED\|Error" $OUT/pytest.log | head
for i in 1 2; do
  python bench.py --quick --steps 6 --warmup 2 > $OUT/head_base_$i.json 2> $OUT/head_base_$i.err
  MDM_HIP_LIB=$V python bench.py --quick --steps 6 --warmup 2 > $OUT/head_epis_$i.json 2> $OUT/head_epis_$i.err
  python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_base_$i.json 2> $OUT/dip_base_$i.err
  MDM_HIP_LIB=$V python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dip_epis_$i.json 2> $OUT/dip_epis_$i.err
  for b in 1 6 10; do
    python bench.py --quick --batch $b --steps 5 --warmup 2 > $OUT/b${b}_base_$i.json 2> $OUT/b${b}_base_$i.err
    MDM_HIP_LIB=$V python bench.py --quick --batch $b --steps 5 --warmup 2 > $OUT/b${b}_epis_$i.json 2> $OUT/b${b}_epis_$i.err
  done
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_*_[12].json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-300:])
PY

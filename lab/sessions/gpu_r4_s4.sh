#!/bin/bash
# Round 4, session 4: (i) the operand split by v_fma_mix (bit-exactness test + same-box headline A/B against the three-instruction
# form), (ii) non-temporal K / V^T streaming and plane stores in the attention kernel (kernel-only probe + whole-bench A/B).
set -u
OUT=gpurun_out/${1:-r4s4}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -k "split or small_batch or clip50" > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"
python tools/attn_probe.py 20 0,64 > $OUT/attn_probe.txt 2>&1; cat $OUT/attn_probe.txt
PROBE=motion-diffusion-model_amd/csrc/libmdm_hip_probe.so
for R in 1 2; do
  for V in "mix:" "nomix:MDM_HIP_LIB=build/variants/libmdm_nomix.so" "probe0:MDM_HIP_LIB=$PROBE" "ntld:MDM_HIP_LIB=$PROBE,MDM_AX_ABL=64" "ntst:MDM_HIP_LIB=$PROBE,MDM_AX_ABL=128" "ntboth:MDM_HIP_LIB=$PROBE,MDM_AX_ABL=192"; do
    TAG=${V%%:*}; ENVS=$(echo "${V#*:}" | tr ',' ' ')
    env $ENVS timeout 300 python bench.py --steps 3 --warmup 1 --quick > $OUT/ab_${TAG}_$R.json 2> $OUT/ab_${TAG}_$R.err
    python - $OUT/ab_${TAG}_$R.json $TAG $R <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("ab", sys.argv[2], "round", sys.argv[3], d["value"], "motions/s", d["kernel_ms"])
except Exception as e:
    print("ab", sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done

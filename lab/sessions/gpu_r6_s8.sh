#!/bin/bash
# Round 6, session 8: condition token written by the pose-transpose kernel (one launch fewer per encoder step).  Encoder parity tests on
# the new build, then same-box A/B against HEAD's csrc (build/variants/libmdm_hip_before.so): latency regime (B = 1 / 6 / 10, 50-step
# loop) and headline, interleaved.
set -u
TAG=${1:-r6s8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -q -x > $OUT/pytest_enc.log 2>&1; tail -2 $OUT/pytest_enc.log
cat > $OUT/lat.py <<'PY'
import json, sys, time, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import make_pair, synth_state_dict, synth_y, to_dev
DEV = "cuda:0"
model, diffusion = make_pair(synth_state_dict(seed=0), 50, DEV, guided=True)
diffusion.check_finite = False
res = {}
for B, n in ((1, 20), (6, 12), (10, 10), (128, 3)):
    y = to_dev(synth_y(B, 196, seed=3), DEV)
    f = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 196), clip_denoised=False, model_kwargs={"y": y}, seed=5)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); res[f"B{B}"] = round((time.perf_counter() - t0) / n * 1e3, 3)
print(json.dumps(res))
PY
for i in 1 2 3 4; do
  for v in before after; do
    if [ $v = after ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_before.so; fi
    python $OUT/lat.py $R > $OUT/lat_${v}_$i.json 2> $OUT/lat_${v}_$i.err
  done
done
unset MDM_HIP_LIB
python - $OUT <<'PY'
import json, sys, glob, statistics as st
out = sys.argv[1]
r = {}
for v in ("before", "after"):
    runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(out + f"/lat_{v}_*.json"))]
    r[v] = {k: [x[k] for x in runs] for k in runs[0]}
print(json.dumps(r))
print({k: round(st.median(r["after"][k]) / st.median(r["before"][k]), 4) for k in r["before"]})
PY

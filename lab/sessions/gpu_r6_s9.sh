#!/bin/bash
# Round 6, session 9: does the HIP runtime's kernel-argument placement matter for the launch-latency-bound loops?
# HIP_FORCE_DEV_KERNARG unset / 0 / 1 (a HIP runtime variable, not one of this library's): DiP B = 32 + per-call, encoder B = 1 / 6.
set -u
TAG=${1:-r6s9}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cat > $OUT/lat.py <<'PY'
import json, sys, time, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import make_pair, synth_state_dict, synth_y, to_dev
DEV = "cuda:0"
model, diffusion = make_pair(synth_state_dict(seed=0), 50, DEV, guided=True)
diffusion.check_finite = False
res = {}
for B, n in ((1, 20), (6, 12)):
    y = to_dev(synth_y(B, 196, seed=3), DEV)
    f = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 196), clip_denoised=False, model_kwargs={"y": y}, seed=5)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); res[f"B{B}"] = round((time.perf_counter() - t0) / n * 1e3, 3)
print(json.dumps(res))
PY
python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > /dev/null 2>&1
for i in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    python bench_dip.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/dip_${v}_$i.json 2> $OUT/dip_${v}_$i.err
    python $OUT/lat.py $R > $OUT/lat_${v}_$i.json 2> $OUT/lat_${v}_$i.err
  done
done
unset HIP_FORCE_DEV_KERNARG
python - $OUT <<'PY'
import json, sys, glob
out = sys.argv[1]
for v in ("unset", "0", "1"):
    for i in (1, 2):
        d = json.load(open(f"{out}/dip_{v}_{i}.json")); l = json.loads(open(f"{out}/lat_{v}_{i}.json").read().strip().splitlines()[-1])
        print(v, i, "dip", d["value"], "call B1/B6", d["small_batch"]["B1"]["window_call_ms"], d["small_batch"]["B6"]["window_call_ms"], "enc", l)
PY

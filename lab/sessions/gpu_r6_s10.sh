#!/bin/bash
# Round 6, session 10: MDM_EARLY_KERNARGS=1 (operand pointers' scalar loads pinned into the entry block of gemm_x3s_kernel /
# selfattn_block_kernel: one scalar-memory round trip instead of two in front of the first LDS-DMA request) vs the product build.
set -u
TAG=${1:-r6s10}
V=${2:-early}
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
for B, n in ((1, 20), (6, 12), (10, 10)):
    y = to_dev(synth_y(B, 196, seed=3), DEV)
    f = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 196), clip_denoised=False, model_kwargs={"y": y}, seed=5)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); res[f"B{B}"] = round((time.perf_counter() - t0) / n * 1e3, 3)
print(json.dumps(res))
PY
python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > /dev/null 2>&1
for i in 1 2 3 4 5; do
  for v in before after; do
    if [ $v = before ]; then unset MDM_HIP_LIB; else export MDM_HIP_LIB=$R/build/variants/libmdm_hip_$V.so; fi
    python bench_dip.py --steps 10 --warmup 2 --no-cpu-baseline $([ $i = 1 ] || echo --no-small-batch) > $OUT/dip_${v}_$i.json 2> $OUT/dip_${v}_$i.err
    if [ $i -le 3 ]; then python $OUT/lat.py $R > $OUT/lat_${v}_$i.json 2> $OUT/lat_${v}_$i.err; fi
  done
done
unset MDM_HIP_LIB
python - $OUT <<'PY'
import json, sys, glob, statistics as st
out = sys.argv[1]
r = {v: [json.load(open(f))["value"] for f in sorted(glob.glob(out + f"/dip_{v}_*.json"))] for v in ("before", "after")}
print(r, "dip ratio of medians", round(st.median(r["after"]) / st.median(r["before"]), 4))
for v in ("before", "after"):
    print(v, "per call", {k: x["window_call_ms"] for k, x in json.load(open(out + f"/dip_{v}_1.json"))["small_batch"].items() if k.startswith("B")})
l = {v: [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(out + f"/lat_{v}_*.json"))] for v in ("before", "after")}
print({k: (round(st.median([x[k] for x in l["before"]]), 3), round(st.median([x[k] for x in l["after"]]), 3)) for k in l["before"][0]})
PY

#!/bin/bash
# Round 5: DiP's remaining GEMMs (out_proj kinds, linear1 / linear2) on 32-row tiles (480 workgroups, two per CU) against the default 64-row tiles (240)
set -u
TAG=${1:-r5diprt}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do
  for rt in 0 1; do
    python bench_dip.py --steps 5 --warmup 2 --no-cpu-baseline --row-tiles $rt > $OUT/dip_rt${rt}_$i.json 2> $OUT/dip_rt${rt}_$i.err
    python bench_dip.py --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --row-tiles $rt > $OUT/dip64_rt${rt}_$i.json 2> $OUT/dip64_rt${rt}_$i.err
  done
done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]["linear"])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-300:])
PY

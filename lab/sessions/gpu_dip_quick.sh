#!/bin/bash
# DiP quick session: the trans_dec GPU parity subset, then bench_dip.py (optionally a second time with the step-by-step loop).
# Usage: bash tools/gpu_dip_quick.sh <tag> [stepwise]
set -u
TAG=${1:-dipq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "dip or dec" > $OUT/pytest_dip.log 2>&1
tail -3 $OUT/pytest_dip.log
for r in 1 2; do
  timeout 300 python bench_dip.py --steps 4 --no-cpu-baseline > $OUT/dip_$r.json 2> $OUT/dip_$r.err
  python -c "
import json; d = json.load(open('$OUT/dip_$r.json')); print('native  ', d['value'], d['ms_per_step'], d['kernel_ms'], d['launches_per_motion_batch'])"
  if [ "${2:-}" = "stepwise" ]; then
    MDM_DIP_STEPWISE=1 timeout 300 python bench_dip.py --steps 4 --no-cpu-baseline > $OUT/dip_step_$r.json 2> $OUT/dip_step_$r.err
    python -c "
import json; d = json.load(open('$OUT/dip_step_$r.json')); print('stepwise', d['value'], d['ms_per_step'], d['kernel_ms'], d['launches_per_motion_batch'])"
  fi
done

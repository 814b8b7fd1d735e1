#!/bin/bash
# Same-box A/B of whole-bench throughput between builds of the library (MDM_HIP_LIB selects the .so), interleaved rounds.
# Usage: bash tools/gpu_ab.sh <tag> <rounds> <lib> [<lib> ...]      ("default" = csrc/libmdm_hip.so;
#        "env:NAME=VALUE" = the default library with that environment variable set, e.g. env:MDM_ENC_GROUPS=2)
# PARITY_K="<pytest -k expr>": first run that GPU parity subset against every non-default library.
set -u
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" != "default" ] && [ "${L#env:}" = "$L" ] && [ -n "${PARITY_K:-}" ]; then
    MDM_HIP_LIB=$PWD/$L timeout 600 python -m pytest tests -m gpu -x -q -k "$PARITY_K" > $OUT/pytest_$(basename $L .so).log 2>&1
    echo "parity $L: $(tail -1 $OUT/pytest_$(basename $L .so).log)"
  fi
done
for r in $(seq 1 $ROUNDS); do
  for L in "$@"; do
    N=$(basename $L .so)
    EV=""
    if [ "$L" = "default" ]; then unset MDM_HIP_LIB
    elif [ "${L#env:}" != "$L" ]; then unset MDM_HIP_LIB; EV="${L#env:}"; N=$(echo "$EV" | tr '=' '_')
    else export MDM_HIP_LIB=$PWD/$L; fi
    env $EV timeout 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${N}_$r.json 2> $OUT/bench_${N}_$r.err
    python - $OUT/bench_${N}_$r.json $N $r <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:16s} round {sys.argv[3]}: {d['value']:8.2f} {d['unit']}  kernel_ms {d['kernel_ms']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
unset MDM_HIP_LIB

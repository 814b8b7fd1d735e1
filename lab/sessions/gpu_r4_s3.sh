#!/bin/bash
# Round 4, session 3: parity of the final small-batch path (both GEMM kernels), smoke, the bench line with its new sub-records.
set -u
OUT=gpurun_out/${1:-r4s3}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "not config1 and not config2 and not f16f6" > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"
grep "parity\]" $OUT/pytest.log > $OUT/parity_lines.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["small_batch"])
PY

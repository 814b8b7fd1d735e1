#!/bin/bash
# Round 6, session 2: the WHOLE GPU suite on the build with (a) the capture-safe dynamic-LDS opt-in, (b) counted asm stores on the
# attention DIRECT path, (c) the X3 tile pin of gemm_f32.h, (d) the long-sequence route (attention_long.h) -- common.h changed, so the
# whole suite runs -- then the markers: headline (quick), DiP with the new per-call latency sub-record.
set -u
TAG=${1:-r6s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
timeout 1700 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest gpu: $(tail -1 $OUT/pytest_gpu.log)"; grep "FAILED\|Error" $OUT/pytest_gpu.log | head -20
grep -o "\[parity\].*" $OUT/pytest_gpu.log | sort -u > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt; grep "T = 400\|dynamic\|shards\|DiP 20" $OUT/parity_lines.txt
python bench.py --quick > $OUT/bench_quick.json 2> $OUT/bench_quick.err; python -c "
import json; d = json.load(open('$OUT/bench_quick.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench_dip.py --steps 3 --warmup 1 > $OUT/dip.json 2> $OUT/dip.err; python -c "
import json; d = json.load(open('$OUT/dip.json')); print('dip', d['value'], d['ms_per_step'], d['launches_per_motion_batch']); print(d.get('small_batch')); print(d.get('cpu_baseline'))"

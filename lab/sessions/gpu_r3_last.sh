#!/bin/bash
set -u
OUT=gpurun_out/r3last
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s -k "config1 or config2 or config4 or hostile or holes or golden_loops or shard_invariance or dip_autoregressive_matches" > $OUT/pytest_s.log 2>&1
echo "pytest subset: $(tail -1 $OUT/pytest_s.log)"
grep "parity\]" $OUT/pytest_s.log > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -c 900 $OUT/bench_full.json

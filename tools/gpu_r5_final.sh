#!/bin/bash
# Round 5, closing session on the final sources: smoke; kernel trace + PMC passes of the headline bench command and of the DiP bench
# (-> profiles/r05_pmc.json, profiles/r05_dip_pmc.json, written here so that the bench line below can quote them); the full bench
# line; the whole GPU suite.
set -u
TAG=${1:-r5final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/gpu_prof.sh $TAG/prof pmc > $OUT/prof.log 2>&1
head -12 $OUT/prof/kernel_stats.md | cut -c1-170
python tools/pmc_to_json.py $OUT/prof profiles/r05_pmc.json > $OUT/pmc_to_json.log 2>&1; cp profiles/r05_pmc.json $OUT/r05_pmc.json
bash tools/gpu_r5_dip_pmc.sh $TAG/dippmc > $OUT/dippmc.log 2>&1
python tools/dip_pmc_to_json.py $OUT/dippmc profiles/r05_dip_pmc.json > $OUT/dip_pmc_to_json.log 2>&1; cp profiles/r05_dip_pmc.json $OUT/r05_dip_pmc.json
head -8 $OUT/dippmc/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["steps1000"]["value"], d["f32_mode"]["value"],
      d["dip"]["value"], d["dip"]["roofline"]["traffic"], d["dip"]["launches_per_motion_batch"], d["cpu_baseline"]["value"], d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"])
PY
timeout 1100 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest gpu: $(tail -1 $OUT/pytest_gpu.log)"; grep "FAILED\|Error" $OUT/pytest_gpu.log | head
grep -o "\[parity\].*" $OUT/pytest_gpu.log | sort -u > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt

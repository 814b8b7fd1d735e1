#!/bin/bash
# Round-4 closing session on the final sources: full GPU suite (parity lines kept), smoke, kernel trace + PMC passes of the bench
# command, the bench line.
set -u
OUT=gpurun_out/${1:-r4final}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
timeout 2400 python -m pytest tests -m gpu -x -q -s --durations=25 > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/gpu_prof.sh ${1:-r4final}/prof pmc > $OUT/prof.log 2>&1
head -14 $OUT/prof/kernel_stats.md | cut -c1-170
for B in 32 128; do for R in planes skeleton; do
  E=A=1; [ $R = skeleton ] && E=MDM_X3S_MAX_SEQS=0
  env $E timeout 300 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > $OUT/dip_${R}_B$B.json 2> $OUT/dip_${R}_B$B.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/dip_${R}_B$B.json
done; done
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"])
PY

#!/bin/bash
# Round 4: gemm_x3s residual tile by LDS-DMA (RES_LDS) -- same-box A/B against the previous product library (build/libmdm_hip_old.so),
# then the closing evidence on the new build: kernel trace + PMC passes, full bench line, GPU suite.
set -u
TAG=${1:-r4resl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
OLD=$PWD/build/libmdm_hip_old.so
ab() {  # tag, command...
  local tag=$1; shift
  for v in new old new old; do
    if [ $v = old ]; then E="MDM_HIP_LIB=$OLD"; else E="A=1"; fi
    env $E timeout 300 "$@" > $OUT/ab_${tag}_$v.json 2>> $OUT/ab.err
    python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['kernel_ms']['linear'])" $OUT/ab_${tag}_$v.json $tag $v
  done
}
ab dip32 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 32
ab dip64 python bench_dip.py --steps 3 --warmup 1 --no-cpu-baseline --batch 64
ab enc10 python bench.py --batch 10 --steps 3 --warmup 1 --quick
ab enc16 python bench.py --batch 16 --steps 3 --warmup 1 --quick
ab enc6 python bench.py --batch 6 --steps 3 --warmup 1 --quick
ab enc1 python bench.py --batch 1 --steps 3 --warmup 1 --quick
timeout 300 python tools/x3s_timeline.py 32 > $OUT/timeline_dip32.txt 2>&1; grep -A5 "launch 7\|launch 11" $OUT/timeline_dip32.txt | head -14
bash tools/gpu_prof.sh $TAG/prof pmc > $OUT/prof.log 2>&1
head -7 $OUT/prof/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"])
PY
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=6 --deselect tests/test_gpu_round2.py::test_config2_B64_T196_1000_steps_replayed_through_the_oracle > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log | sed 's/^[.s]*//' > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt

#!/bin/bash
# Round 3, session 12: epilogue work (clipped windows, v_fma_mix residual, column scale compiled out) -- full GPU suite on the
# final sources, same-box A/B against the two previous builds, then the closing measurements (tools/gpu_r3_final.sh's).
set -u
OUT=gpurun_out/r3s12
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256())" > $OUT/csrc_sha256.txt
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log > $OUT/parity_lines.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
BENCH_ARGS="--no-extras --steps 3" bash tools/gpu_ab.sh r3s12/ab 2 build/libmdm_hip_BASE.so build/libmdm_hip_CLIP.so default 2>&1 | tee $OUT/ab.txt
bash tools/gpu_prof.sh r3s12/prof pmc > $OUT/prof.log 2>&1
head -14 $OUT/prof/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"], d["steps1000"]["value"], d["f32_mode"]["value"], d["dip"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY

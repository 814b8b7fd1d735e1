#!/bin/bash
# Round 4: the GPU suite on the final build hash, minus its four longest tests (they ran on the same product code in r4final3 / r4resl).
set -u
OUT=gpurun_out/${1:-r4suite}
mkdir -p $OUT
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
timeout 225 python -m pytest tests -m gpu -x -q -s \
  --deselect tests/test_gpu_round2.py::test_config2_B64_T196_1000_steps_replayed_through_the_oracle \
  --deselect tests/test_gpu_round2.py::test_config1_B128_T196_50_steps_eight_samples_replayed_through_the_oracle \
  --deselect "tests/test_gpu_round2.py::test_hostile_weights_forward_and_loop[small-f16x3]" \
  --deselect tests/test_gpu_parity.py::test_eval_caller_call_sequence > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest_gpu.log)"
grep "parity\]" $OUT/pytest_gpu.log | sed 's/^[.s]*//' > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt

#!/bin/bash
# Round 6 closing session, GATED (DESIGN section 9.5 used to be prose): the bench line of a build is produced ONLY
# if the WHOLE GPU suite passed on the SAME binary in the SAME session.  Order: identity of the binary -> smoke -> pytest -m gpu (all
# of it) -> gate -> kernel traces + PMC passes (-> profiles/r06_pmc.json, r06_dip_pmc.json, which the bench line quotes) -> bench line.
# Writes $OUT/closing_gate.json = {lib_sha256, csrc_sha256, pytest summary, gate}; profiles/README.md names the pair
# (closing_gate.json, bench line) of the round.  A red suite leaves no bench line behind (exit 3).
set -u
TAG=${1:-r6final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha256()); print(bench.lib_sha256())" > $OUT/csrc_sha256.txt
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
PYRC=$?
SUMMARY=$(tail -1 $OUT/pytest_gpu.log)
echo "pytest gpu (rc $PYRC): $SUMMARY"; grep "FAILED\|Error" $OUT/pytest_gpu.log | head
grep -o "\[parity\].*" $OUT/pytest_gpu.log | sort -u > $OUT/parity_lines.txt; wc -l $OUT/parity_lines.txt
python - "$OUT" "$PYRC" "$SUMMARY" <<'PY'
import json, sys
out, rc, summary = sys.argv[1], int(sys.argv[2]), sys.argv[3]
csrc, lib = open(out + "/csrc_sha256.txt").read().split()[:2]
green = rc == 0 and " passed" in summary and "failed" not in summary and "error" not in summary.lower()
json.dump({"csrc_sha256": csrc, "lib_sha256": lib, "pytest_rc": rc, "pytest_summary": summary.strip(), "gate": "open" if green else "CLOSED"},
          open(out + "/closing_gate.json", "w"), indent=1)
print("closing gate:", "open" if green else "CLOSED")
sys.exit(0 if green else 3)
PY
if [ $? -ne 0 ]; then echo "GPU suite not green on this binary: NO bench line is produced"; exit 3; fi
bash tools/gpu_prof.sh $TAG/prof pmc > $OUT/prof.log 2>&1
head -12 $OUT/prof/kernel_stats.md | cut -c1-170
python tools/pmc_to_json.py $OUT/prof profiles/r06_pmc.json > $OUT/pmc_to_json.log 2>&1; cp profiles/r06_pmc.json $OUT/r06_pmc.json
bash tools/gpu_dip_pmc.sh $TAG/dippmc > $OUT/dippmc.log 2>&1
python tools/dip_pmc_to_json.py $OUT/dippmc profiles/r06_dip_pmc.json > $OUT/dip_pmc_to_json.log 2>&1; cp profiles/r06_dip_pmc.json $OUT/r06_dip_pmc.json
head -8 $OUT/dippmc/kernel_stats.md | cut -c1-170
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
d = json.load(open(out + "/bench_full.json"))
g = json.load(open(out + "/closing_gate.json"))
assert d["build"]["lib_sha256"] == g["lib_sha256"], "the bench ran on another binary than the suite"
g["bench_value"] = d["value"]; g["bench_lib_sha256"] = d["build"]["lib_sha256"]
json.dump(g, open(out + "/closing_gate.json", "w"), indent=1)
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["steps1000"]["value"], d["f32_mode"]["value"],
      d["dip"]["value"], d["dip"]["roofline"]["traffic"], d["dip"]["launches_per_motion_batch"], d["dip"].get("small_batch"), d["cpu_baseline"]["value"],
      d["small_batch"]["B1"], d["small_batch"]["B6"], d["small_batch"]["B10"], d.get("trans_dec"))
PY

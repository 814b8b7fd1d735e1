#!/bin/bash
# Round 6, session 1: (a) the phase-boundary probe of VERDICT r05 item 3 step 0 (lab/probes/phase_barrier); (b) round 6's new GPU
# tests (dynamic text vs the reference fixture, DiP shard invariance bitwise); (c) DiP kernel trace with inter-kernel gaps; (d) DiP
# bench marker.
set -u
TAG=${1:-r6s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 lab/probes/phase_barrier/phase_barrier 2000 > $OUT/phase_barrier.jsonl 2> $OUT/phase_barrier.err
cat $OUT/phase_barrier.jsonl | cut -c1-260; tail -3 $OUT/phase_barrier.err
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x > $OUT/pytest_r6.log 2>&1
echo "pytest r6: $(tail -1 $OUT/pytest_r6.log)"; grep "FAILED\|Error\|parity" $OUT/pytest_r6.log | head -20
bash tools/gpu_dip_trace.sh $TAG/diptrace > $OUT/diptrace.log 2>&1
cat $OUT/diptrace/kernel_gaps.md | cut -c1-220
head -10 $OUT/diptrace/kernel_stats.md | cut -c1-200

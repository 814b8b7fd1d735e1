#!/bin/bash
# Round 6, session 3: (a) why the capture test did not refuse; (b) eager vs hipGraph replay of the loops (device-side gaps);
# (c) round-6 tests on the build after the mdm_api.hip split (device code object byte-identical to the pre-split build).
set -u
TAG=${1:-r6s3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python lab/probes/dbg_capture.py > $OUT/dbg_capture.log 2>&1; tail -6 $OUT/dbg_capture.log
timeout 600 python lab/probes/graph_replay_probe.py > $OUT/graph_replay.jsonl 2> $OUT/graph_replay.err; cat $OUT/graph_replay.jsonl; tail -3 $OUT/graph_replay.err
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q -x -k "not first_use" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log

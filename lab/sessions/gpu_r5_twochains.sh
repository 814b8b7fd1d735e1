#!/bin/bash
# Round 5: the headline batch as two concurrent half-batch chains sharing the chip (lab/probes/two_chains.py; probe library hooks
# MDM_CHAIN_FREE / MDM_X3_GRID_DIV) against one chain.
set -u
TAG=${1:-r5two}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/motion-diffusion-model_amd/csrc/libmdm_hip_probe.so
python bench.py --quick --steps 3 --warmup 1 > $OUT/marker.json 2> $OUT/marker.err
for i in 1 2; do
  MDM_HIP_LIB=$P TWO_CHAINS_MODE=one timeout 120 python lab/probes/two_chains.py 3 > $OUT/one_$i.json 2> $OUT/one_$i.err
  MDM_HIP_LIB=$P MDM_CHAIN_FREE=1 MDM_X3_GRID_DIV=2 TWO_CHAINS_MODE=par timeout 120 python lab/probes/two_chains.py 3 > $OUT/par_div2_$i.json 2> $OUT/par_div2_$i.err
  MDM_HIP_LIB=$P MDM_CHAIN_FREE=1 TWO_CHAINS_MODE=par timeout 120 python lab/probes/two_chains.py 3 > $OUT/par_div1_$i.json 2> $OUT/par_div1_$i.err
done
MDM_HIP_LIB=$P MDM_CHAIN_FREE=1 MDM_X3_GRID_DIV=2 TWO_CHAINS_MODE=seq timeout 200 python lab/probes/two_chains.py 2 > $OUT/seq_par_div2.json 2> $OUT/seq_par_div2.err
for f in $OUT/*.json; do echo "$(basename $f): $(tail -1 $f | cut -c1-420)"; done
tail -3 $OUT/*.err | grep -v "^$" | grep -i "error\|Traceback\|assert" | head

#!/bin/bash
# The four soaks of tools/stress.py (random dictionaries and texts against the oracle) in one gpurun call: tools/soak_round.sh r05 [seconds each] [seed]
TAG=${1:-rXX}; T=${2:-60}; SEED=${3:-51}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
: > $OUT/${TAG}_soak.txt
for kind in engines gram select iter; do
  timeout $((T * 3 + 120)) python $R/tools/stress.py $T $SEED $kind 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/${TAG}_soak.txt
done
cat $OUT/${TAG}_soak.txt

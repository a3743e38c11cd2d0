#!/bin/bash
# randomised soak of the round's new paths on the GPU box: tools/stress.py <seconds> <seed> engines (several seeds)
cd $GRAFT_REPO_ROOT
for seed in 41 42 43; do
  timeout 200 python tools/stress.py 75 $seed engines 2>&1 | grep -v amdgpu.ids | tail -4
done > gpurun_out/r04_soak.txt 2>&1
cat gpurun_out/r04_soak.txt

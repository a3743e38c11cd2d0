#!/bin/bash
# randomised soak of the round's new paths on the GPU box: tools/stress.py <seconds> <seed> engines | select (two seeds each)
cd $GRAFT_REPO_ROOT
{
for seed in 41 42; do timeout 200 python tools/stress.py 60 $seed engines 2>&1 | grep -v amdgpu.ids | tail -4; done
for seed in 51 52; do timeout 200 python tools/stress.py 60 $seed select 2>&1 | grep -v amdgpu.ids | tail -4; done
} > gpurun_out/r04_soak.txt 2>&1
cat gpurun_out/r04_soak.txt

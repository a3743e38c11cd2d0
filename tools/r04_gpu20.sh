#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/ab_wide.py 256 > gpurun_out/r04_c20_ab_wide.txt 2>&1; cat gpurun_out/r04_c20_ab_wide.txt | grep -v amdgpu.ids

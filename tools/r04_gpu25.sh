#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pfx.py -x -q -m gpu > gpurun_out/r04_c25_pytest.log 2>&1; tail -5 gpurun_out/r04_c25_pytest.log
timeout 600 python tools/ab_wide.py 1024 > gpurun_out/r04_c25_ab_wide.txt 2>&1; grep -v "amdgpu.ids\|Tiered" gpurun_out/r04_c25_ab_wide.txt | cut -c1-140

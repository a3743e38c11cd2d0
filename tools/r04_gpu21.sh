#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pfx.py -x -q -m gpu > gpurun_out/r04_c21_pytest.log 2>&1; tail -12 gpurun_out/r04_c21_pytest.log

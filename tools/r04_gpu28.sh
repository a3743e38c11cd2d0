#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "at_size" > gpurun_out/r04_c28_pytest.log 2>&1; tail -15 gpurun_out/r04_c28_pytest.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r04_c22_pytest_full.log 2>&1; tail -25 gpurun_out/r04_c22_pytest_full.log

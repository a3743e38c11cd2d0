#!/bin/bash
# PFX tuple emission: parity tests, then rates
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pfx.py -x -q -m gpu > gpurun_out/r04_c18_pytest.log 2>&1; tail -15 gpurun_out/r04_c18_pytest.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compact or lazy or concurrent or iterator or facade or native" > gpurun_out/r04_c26_pytest.log 2>&1; tail -5 gpurun_out/r04_c26_pytest.log
timeout 600 python tools/time_iter.py 1024 > gpurun_out/r04_c26_iter.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_c26_iter.txt

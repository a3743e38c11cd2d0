#!/bin/bash
# the round's last word on the GPU box: the whole -m gpu suite, smoke(), then the randomised soaks (tools/r04_soak.sh)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu > gpurun_out/r04_full_gpu_suite.log 2>&1; tail -4 gpurun_out/r04_full_gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04_smoke.txt
bash tools/r04_soak.sh

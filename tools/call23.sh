#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_soak.py -x -q -m gpu -k "gram or cfg3 or soak or wide" 2>&1 | tail -3
bash tools/call19.sh

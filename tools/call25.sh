#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "step_back or tail_records" 2>&1 | tail -15

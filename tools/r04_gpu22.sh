#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_c22_smoke.log 2>&1; tail -2 gpurun_out/r04_c22_smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r04_c22_pytest_full.log 2>&1; tail -14 gpurun_out/r04_c22_pytest_full.log

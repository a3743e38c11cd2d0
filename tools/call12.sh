#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest $R/tests/test_gpu_charwise.py $R/tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for m in leftmost find overlapping; do
    timeout 200 python tools/bench_cfg5.py --mode $m --cpu-mib 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['match_count'])"
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python $R/tools/time_emit.py 1024 sparse 3 2>&1 | grep "emit="
timeout 200 python tools/bench_cfg5.py --mode leftmost --cpu-mib 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 leftmost', d['value'], d['ms_per_step'], d['match_count'])"
timeout 200 python tools/time_find.py 1024 sparse 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2

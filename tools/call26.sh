#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -3
for o in 0 1; do timeout 200 python tools/bench_cfg5.py --mode overlapping --cpu-mib 4 --opt overlap_micro=$o 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 overlapping micro=$o', d['value'], d['ms_per_step'], d['match_count'], d['cpu_baseline']['parity_with_gpu_on_sample'])"; done
for h in sparse dense; do
python tools/sweep.py --workload cfg3 --haystack $h --mib 1024 --reps 3 --grid "engine=darray,tiered;overlap_micro=0,2" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$h" '{print w, $0}'
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_soak.py -x -q -m gpu -k "gram or cfg3 or soak or wide" 2>&1 | tail -3
for rep in 1 2; do for h in sparse dense; do
  python tools/sweep.py --workload cfg3 --haystack $h --mib 4096 --reps 10 --grid "engine=gram;gram_version=2;count_only=1" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' -e "s/{.*}//" | awk -v w="$h" '{print w, $0}'
done; done

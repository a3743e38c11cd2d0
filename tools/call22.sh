#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_soak.py -x -q -m gpu -k "gram or cfg3 or soak or wide or emit" 2>&1 | tail -3
for h in sparse dense; do
  python tools/sweep.py --workload cfg3 --haystack $h --mib 4096 --reps 10 --grid "engine=gram;gram_version=1,2;count_only=0" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$h" '{print w, $0}'
done
timeout 300 python $R/tools/time_emit.py 1024 sparse 3 2>&1 | grep "emit=1"
timeout 300 python $R/tools/time_emit.py 1024 dense 3 2>&1 | grep "emit=1"

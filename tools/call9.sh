#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest $R/tests/test_gpu_charwise.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for m in leftmost find; do
  for o in 0 1; do
    echo "mode=$m char_map_lds=$o"; timeout 200 python $R/tools/bench_cfg5.py --mode $m --cpu-mib 4 --opt char_map_lds=$o 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['match_count'], d.get('cpu_baseline',{}).get('parity_with_gpu_on_sample'))"
  done
done
timeout 200 python $R/tools/time_find.py 1024 sparse 2>&1 | grep -v "^W\|^E" | tail -3

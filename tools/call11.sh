#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest $R/tests/test_gpu_charwise.py $R/tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in /tmp/_orig.so abtmp/lib_cw_nomicro.so; do
  [ $f != /tmp/_orig.so ] && cp $f $L; echo "== $f (cfg5)"
  for m in leftmost find; do for o in 0 1; do
    timeout 200 python tools/bench_cfg5.py --mode $m --cpu-mib 0 --opt char_map_lds=$o 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m maplds=$o', d['value'], d['ms_per_step'], d['match_count'])"
  done; done
done
cp /tmp/_orig.so $L

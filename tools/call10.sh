#!/bin/bash
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in abtmp/lib_rs_*.so; do
  cp $f $L; echo "== $f"
  timeout 200 python tools/time_find.py 1024 sparse 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2
done
for f in abtmp/lib_rs_base.so abtmp/lib_cw_*.so; do
  cp $f $L; echo "== $f (cfg5)"
  for m in leftmost find; do for o in 0 1; do
    timeout 200 python tools/bench_cfg5.py --mode $m --cpu-mib 0 --opt char_map_lds=$o 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m maplds=$o', d['value'], d['ms_per_step'], d['match_count'])"
  done; done
done
cp /tmp/_orig.so $L

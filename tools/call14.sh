#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_steppers.py $R/tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in /tmp/_orig.so abtmp/lib_rs_nomicro.so; do
  [ $f != /tmp/_orig.so ] && cp $f $L; echo "== $f"
  for w in sparse dense; do timeout 200 python tools/time_find.py 1024 $w 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2; done
done
cp /tmp/_orig.so $L

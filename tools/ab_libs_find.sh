#!/bin/bash
# tools/time_find.py for the shipped library and then for every abtmp/lib_<name>.so
L=daachorse_amd/lib/libdaachorse_amd.so
M=${1:-1024}
for hk in sparse dense; do echo "== shipped $hk"; python tools/time_find.py $M $hk 2>&1 | grep -v amdgpu.ids; done
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  for hk in sparse dense; do echo "== $f $hk"; python tools/time_find.py $M $hk 2>&1 | grep -v amdgpu.ids; done
done
cp /tmp/_orig.so $L

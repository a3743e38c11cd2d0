#!/bin/bash
# ab_count.py for the shipped library and then for every abtmp/lib_<name>.so (timing experiments; their counts may be wrong on purpose)
L=daachorse_amd/lib/libdaachorse_amd.so
M=${1:-4096}; V=${2:-tools/ab_count_shapes.json}
echo "== shipped"; python tools/ab_count.py $M $V 2>&1 | grep -v amdgpu.ids
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  echo "== $f"; python tools/ab_count.py $M $V 2>&1 | grep -v amdgpu.ids
done
cp /tmp/_orig.so $L

#!/bin/bash
# A/B of library variants on one box: for every abtmp/lib_<name>.so, GRAM GB/s on cfg3 sparse (and cfg2 with "all")
# usage (through gpurun): bash tools/ab_libs.sh [all]
L=daachorse_amd/lib/libdaachorse_amd.so
ALL=${1:-}
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  for w in "cfg3 sparse 2048 10" ${ALL:+"cfg3 dense 2048 5" "cfg2 sparse 1024 20"}; do
    set -- $w
    python tools/sweep.py --workload $1 --haystack $2 --mib $3 --reps $4 --grid "engine=gram,gram" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$f $1/$2" '{print w, $3, $4, $5, $6, $9, $10}'
  done
done
cp /tmp/_orig.so $L

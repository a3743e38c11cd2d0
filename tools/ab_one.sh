#!/bin/bash
# A/B of abtmp/lib_*.so on ONE workload: bash tools/ab_one.sh cfg3 dense 1024 3
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  python tools/sweep.py --workload $1 --haystack $2 --mib $3 --reps $4 --grid "engine=gram" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$f $1/$2" '{print w, $3, $4, $5, $6, $8, $9}'
done
cp /tmp/_orig.so $L

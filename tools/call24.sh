#!/bin/bash
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in abtmp/lib_g2_nt.so abtmp/lib_g2_cached.so; do
cp $f $L
for h in dense sparse; do
  python tools/sweep.py --workload cfg3 --haystack $h --mib 4096 --reps 10 --grid "engine=gram;gram_version=2;count_only=1;gram_slab=2048,1280" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$f $h" '{print w, $0}'
done; done
cp /tmp/_orig.so $L

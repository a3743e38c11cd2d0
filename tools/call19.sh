#!/bin/bash
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for rep in 1; do
for f in /tmp/_orig.so; do
  cp $f $L
  for h in sparse dense; do
  python tools/sweep.py --workload cfg3 --haystack $h --mib 4096 --reps 10 --grid "engine=gram;gram_version=2;count_only=1" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' -e "s/{.*}//" | awk -v w="$f $h" '{print w, $0}'
  done
done
done
cp /tmp/_orig.so $L

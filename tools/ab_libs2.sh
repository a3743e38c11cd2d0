#!/bin/bash
# A/B of library variants on one box: for every abtmp/lib_<name>.so the sweep grid given as $1 (default: v2 count-only on cfg3 sparse)
# usage (through gpurun): bash tools/ab_libs2.sh ["grid"] [mib] [haystack]
L=daachorse_amd/lib/libdaachorse_amd.so
G=${1:-"engine=gram;gram_version=2;count_only=1"}
M=${2:-2048}
H=${3:-sparse}
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  python tools/sweep.py --workload cfg3 --haystack $H --mib $M --reps 8 --grid "$G" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$f" '{print w, $0}'
done
cp /tmp/_orig.so $L

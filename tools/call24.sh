#!/bin/bash
for h in dense sparse; do
  python tools/sweep.py --workload cfg3 --haystack $h --mib 4096 --reps 10 --grid "engine=gram;gram_version=2;count_only=1;gram_region=0,8192,16384,32768,131072" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$h" '{print w, $0}'
done

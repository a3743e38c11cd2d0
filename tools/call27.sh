#!/bin/bash
for h in sparse dense; do
python tools/sweep.py --workload cfg3 --haystack $h --mib 1024 --reps 3 --grid "engine=darray,tiered;overlap_micro=0,2" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$h" '{print w, $0}'
done
python tools/sweep.py --workload cfg2 --haystack sparse --mib 1024 --reps 3 --grid "engine=darray,tiered;overlap_micro=0,2" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="cfg2" '{print w, $0}'

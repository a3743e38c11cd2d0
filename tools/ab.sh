#!/bin/bash
# quick A/B numbers on one box: GRAM GB/s for cfg3 sparse / dense and cfg2 (each configuration measured twice)
# usage (through gpurun): bash tools/ab.sh [extra grid, e.g. "gram_ppl=16,32"]
G=${1:-engine=gram,gram}
for w in "cfg3 sparse 2048 10" "cfg3 dense 2048 5" "cfg2 sparse 1024 20"; do
  set -- $w
  python tools/sweep.py --workload $1 --haystack $2 --mib $3 --reps $4 --grid "$G" 2>&1 | grep -v amdgpu.ids | sed -e 's/NA=.*lds=[0-9]*//' | awk -v w="$1/$2" '{print w, $0}'
done

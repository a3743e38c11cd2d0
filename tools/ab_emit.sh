#!/bin/bash
# A/B of library variants of the tuple emitter: for every abtmp/lib_<name>.so, tools/time_emit.py (through gpurun)
L=daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do
  cp $f $L
  python tools/time_emit.py ${1:-1024} ${2:-sparse} 3 2>&1 | grep "emit=1" | awk -v w="$f" '{print w, $0}'
done
cp /tmp/_orig.so $L

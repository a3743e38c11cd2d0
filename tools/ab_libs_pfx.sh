#!/bin/bash
L=daachorse_amd/lib/libdaachorse_amd.so
echo "== shipped"; python tools/ab_pfx.py ${1:-1024} 2>&1 | grep -v "amdgpu.ids\|darray"
cp $L /tmp/_orig.so
for f in abtmp/lib_*.so; do cp $f $L; echo "== $f"; python tools/ab_pfx.py ${1:-1024} 2>&1 | grep -v "amdgpu.ids\|darray"; done
cp /tmp/_orig.so $L

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
for f in $R/abtmp/lib_e_*.so; do
  cp $f $L; echo "== $f"
  rm -rf /tmp/p15; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p15 -o e -- python $R/tools/time_emit.py 1024 sparse 2 2>&1 | grep "emit=1"
  find /tmp/p15 -name '*kernel_stats.csv' -exec cat {} \; | grep "emit_kernel" | cut -d, -f1-4 | sed -e 's/daac::Gram2EmitDev.*)"/"/' 
done
cp /tmp/_orig.so $L

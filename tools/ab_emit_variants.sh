#!/bin/bash
# Times the shipped library and every abtmp/lib_e3_*.so (timing-only variants of emit3_kernels.hip built with
# SRC=emit3_kernels tools/mkvar2.sh e3_<NAME> -DE3X_<NAME>, tools/variants/emit3_decomposition.patch) on 1 GiB of cfg3, 16-byte tuples, with the
# kernel times of each (profiles/r04_emit3_experiments.txt).  On the GPU box through gpurun:  bash tools/ab_emit_variants.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd /tmp && export TMPDIR=/tmp
cp $L /tmp/_orig.so
for f in /tmp/_orig.so $R/abtmp/lib_e3_*.so; do
  cp $f $L 2>/dev/null; n=$(basename $f .so); rm -rf /tmp/pe_$n
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$n -o t -- python $R/tools/time_emit.py 1024 sparse 3 0 only16 > /tmp/pe_$n.log 2>&1
  echo "== $n"; grep "emit=1" /tmp/pe_$n.log | cut -c1-120; python $R/tools/kstats.py /tmp/pe_$n emit3_expand,emit3_detect,emit3_bin
done > $O/emit_variants.txt 2>&1
cp /tmp/_orig.so $L
cat $O/emit_variants.txt
